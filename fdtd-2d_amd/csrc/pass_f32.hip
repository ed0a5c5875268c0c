// float32 instantiation of the temporally blocked pass (k_bulk, k_zone, k_pass_pml; passes of
// up to 8 steps -- the 16-step kernels compile in pass_f32_long.hip, in parallel).
#define FDTD_PASS_LONG_EXTERN
#include "pass_impl.hpp"
namespace fdtd_host {
template int launch_pass<float>(fdtd2d *, int, int, int, int, int, const double *, bool, bool, bool, int, int, int);
}
