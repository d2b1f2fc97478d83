#include "eonerf_kernels.h"
hipError_t eo_launch_mlp_bwd(const MlpBwdArgs&, bool, bool, bool, int, hipStream_t) { return hipErrorNotSupported; }
