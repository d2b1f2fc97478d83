#include "eonerf_kernels.h"
hipError_t eo_launch_wgrad(const WgradJob*, int, const int*, int, bool, hipStream_t) { return hipErrorNotSupported; }
