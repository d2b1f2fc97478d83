#include "eonerf_rays.h"
