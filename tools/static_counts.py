#!/usr/bin/env python3
"""Regenerates profiles/static_counts.json from the compiled code object (no GPU needed: hipcc cross-compiles).

    python3 tools/static_counts.py            # writes profiles/static_counts.json, stamped with bench.arithmetic_source_hash()

bench.py quotes `roofline.alu.mad_floor` -- the accumulate kernel's time if it did nothing but the multiply-adds of its bucket
additions -- from this file, and only while the file's hash equals the hash of the current arithmetic sources.  What is recorded:
  * mads_per_madd: the multiply-adds of ONE XYZZ += affine addition as the FORMULA has them (2 fe_mul_minus + 4 fe_mul: 162 each,
    2 fe_sqr: 45 + 81, 1 fe_mulsub: 162 + 81 -- a product is 81 limb products plus 81 for the reduction): 1467;
  * a check of that figure against the code object: the instruction stream of msm_accumulate_kernel<Fp>'s list loop is cut into
    basic blocks, and the multiply-add blocks the no-exception path runs through (the product that gives U2 - X1 ahead of the zero
    tests, and the one big block behind them) must hold exactly that many v_mad_u64_u32;
  * whole-kernel and hot-path instruction counts by mnemonic."""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "interactive-zkp-study_amd", "csrc")
FORMULA_MADS = 2 * 162 + 4 * 162 + 2 * (45 + 81) + (162 + 81)

TU = """#include "msm_impl.h"
namespace zk {
template __global__ void msm_accumulate_kernel<Fp>(const PackedAffine<Fp>*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, Xyzz<Fp>*,
                                                   uint32_t, uint32_t, SortBufs, Xyzz<Fp>*, uint32_t);
}
"""


def compile_kernel():
    with tempfile.TemporaryDirectory(dir=CSRC) as tmp:          # inside csrc/: the headers include ../../include/zkhip.h
        src = os.path.join(tmp, "k.hip")
        open(src, "w").write(TU)
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only", "-Wno-unused-function",
                               "-I" + CSRC, "-S", src, "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_body(asm):
    m = re.search(r"^(_ZN2zk21msm_accumulate_kernelINS_2FeINS_5FpTagE[^:\n]*):.*?\n(.*?)\n\s*s_endpgm", asm, re.S | re.M)
    if not m:
        raise SystemExit("msm_accumulate_kernel<Fp> not found in the assembly")
    return m.group(2).split("\n")


def blocks(lines):
    """[(label, in_loop, [mnemonics])] -- basic blocks in layout order; a block belongs to the list loop when the compiler's comment says
    'in Loop' / 'Loop Header' at depth 1."""
    out, cur, lab, loop = [], [], "entry", False
    for ln in lines:
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            out.append((lab, loop, cur))
            lab, cur, loop = t.split(":")[0], [], False
            continue
        if t.startswith(";"):
            if "Loop" in t and "Depth=1" in t:
                loop = True
            continue
        if not t or t.startswith("."):
            continue
        cur.append(t.split()[0])
    out.append((lab, loop, cur))
    return out


def main():
    asm = compile_kernel()
    bl = blocks(kernel_body(asm))
    whole = collections.Counter(m for _, _, ms in bl for m in ms)
    valu = sum(v for k, v in whole.items() if k.startswith("v_"))
    in_loop = [(lab, ms) for lab, lp, ms in bl if lp]
    # the no-exception path's arithmetic: the blocks of the loop, in layout order up to its largest one, that hold multiply-adds -- the
    # product giving U2 - X1 ahead of the zero tests and everything behind them (the doubling path is laid out after the main block)
    sizes = [len(ms) for _, ms in in_loop]
    last = sizes.index(max(sizes)) if sizes else -1
    path = [(lab, ms) for lab, ms in in_loop[:last + 1] if collections.Counter(ms).get("v_mad_u64_u32", 0)]
    hot = collections.Counter(m for _, ms in path for m in ms)
    hot_valu = {k: v for k, v in hot.items() if k.startswith("v_")}
    hot_mads = hot.get("v_mad_u64_u32", 0)
    sys.path.insert(0, ROOT)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    top = sorted(hot_valu.items(), key=lambda kv: -kv[1])
    rec = {
        "arithmetic_source_sha256": bench.arithmetic_source_hash(),
        "mads_per_madd": FORMULA_MADS,
        "how": "the formula's count: 2 fe_mul_minus (162 each) + 4 fe_mul (162) + 2 fe_sqr (45 + 81) + 1 fe_mulsub (162 + 81); a product is 81 limb "
               "products plus 81 for the reduction.  Checked by tools/static_counts.py against the compiled msm_accumulate_kernel<Fp> (hipcc -S, gfx950): "
               "v_mad_u64_u32 in the multiply-add blocks of the list loop up to its main block (the blocks without multiply-adds on the path -- gather, "
               "unpack, sign, zero tests -- are not in hot_path_vector_instructions; the counter pass gives the dynamic total per addition)",
        "hot_path_blocks": [{"label": lab, "instructions": len(ms), "v_mad_u64_u32": collections.Counter(ms).get("v_mad_u64_u32", 0)} for lab, ms in path],
        "hot_path_mads_in_code_object": hot_mads,
        "formula_matches_code_object": bool(hot_mads == FORMULA_MADS),
        "whole_kernel_static": {"v_mad_u64_u32": whole.get("v_mad_u64_u32", 0), "valu": valu,
                                "note": "includes the heavy-bucket tasks and the doubling / infinity paths that a list entry almost never takes"},
        "hot_path_vector_instructions": dict([("total", sum(hot_valu.values()))] + top[:6] + [("other", sum(v for _, v in top[6:]))]),
        "loop_blocks": [{"label": lab, "instructions": len(ms), "v_mad_u64_u32": collections.Counter(ms).get("v_mad_u64_u32", 0)} for lab, ms in in_loop
                        if len(ms) >= 40],
    }
    path = os.path.join(ROOT, "profiles", "static_counts.json")
    json.dump(rec, open(path, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("mads_per_madd", "hot_path_mads_in_code_object", "formula_matches_code_object", "hot_path_blocks")}))
    print("wrote", path)


if __name__ == "__main__":
    main()
