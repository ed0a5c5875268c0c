// float32 12- and 16-step passes (k_bulk<12>, k_bulk_split<16, 4|8, ...>, k_zone<12|16>).
#include "pass_impl.hpp"
namespace fdtd_host {
template int launch_pass_nt<float, 16>(fdtd2d *, fdtd::PassParams<float> &);
template int launch_pass_nt<float, 12>(fdtd2d *, fdtd::PassParams<float> &);
}
