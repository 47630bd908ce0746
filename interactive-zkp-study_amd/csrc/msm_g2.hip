// msm_g2.hip -- G2 (F_p^2) instantiation of the MSM pipeline.
#include "msm_impl.h"
namespace zk {
MsmPlanBase *msm_plan_new_g2(size_t max_n, bool all_lanes, int chunk_log) { return new MsmPlanImpl<Fp2>(max_n, all_lanes, chunk_log); }
}
