// dac_kernels.h -- launch wrappers of the codec kernels (dac_kernels.hip).
#pragma once
#include "common.h"
