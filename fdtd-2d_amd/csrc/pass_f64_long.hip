// float64 16-step passes (k_bulk_split<double, 16, 4|8, ...>, 2 columns per lane; zone tiles as k_zone<double, 16> with
// 79 KB of dynamic LDS beside the bulk).
#include "pass_impl.hpp"
namespace fdtd_host {
template int launch_pass_nt<double, 16>(fdtd2d *, fdtd::PassParams<double> &);
}
