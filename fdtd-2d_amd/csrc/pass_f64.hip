// float64 instantiation of the temporally blocked pass (k_bulk, k_zone, k_pass_pml).
#define FDTD_PASS_LONG_EXTERN
#include "pass_impl.hpp"
namespace fdtd_host {
template int launch_pass<double>(fdtd2d *, int, int, int, int, int, const double *, bool, bool, bool, int, int, int);
}
