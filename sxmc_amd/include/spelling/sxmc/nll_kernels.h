// <sxmc/nll_kernels.h> as the reference's sources include it (mcmc.h:19).
#pragma once
#include "../hemi/hemi.h"
