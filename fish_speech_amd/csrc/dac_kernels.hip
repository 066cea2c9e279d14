#include "dac_kernels.h"
