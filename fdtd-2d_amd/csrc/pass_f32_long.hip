// float32 16-step passes (k_bulk_split<16, 4|8, ...> with their fused zone tiles).
#include "pass_impl.hpp"
namespace fdtd_host {
template int launch_pass_nt<float, 16>(fdtd2d *, fdtd::PassParams<float> &);
}
