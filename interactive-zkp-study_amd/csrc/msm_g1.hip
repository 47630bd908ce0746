// msm_g1.hip -- G1 (F_p) instantiation of the MSM pipeline.
#include "msm_impl.h"
namespace zk {
MsmPlanBase *msm_plan_new_g1(size_t max_n, bool all_lanes, int chunk_log) { return new MsmPlanImpl<Fp>(max_n, all_lanes, chunk_log); }
}
