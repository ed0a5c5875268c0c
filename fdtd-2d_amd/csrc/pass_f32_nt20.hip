// float32 20-step passes (k_bulk_split<20, 4 waves x 5 levels>; zone tiles as k_zone<20> beside the bulk).
#include "pass_impl.hpp"
namespace fdtd_host {
template int launch_pass_nt<float, 20>(fdtd2d *, fdtd::PassParams<float> &);
}
