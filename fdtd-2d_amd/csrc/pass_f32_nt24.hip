// float32 24-step passes (k_bulk_split<24, 8 waves x 3 levels> with fused zone tiles).
#include "pass_impl.hpp"
namespace fdtd_host {
template int launch_pass_nt<float, 24>(fdtd2d *, fdtd::PassParams<float> &);
}
