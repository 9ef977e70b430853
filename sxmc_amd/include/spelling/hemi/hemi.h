// hemi/hemi.h -- SPELLING ONLY (see hemi/array.h).  mcmc.cpp launches the NLL kernels as
// HEMI_KERNEL_LAUNCH(name, grid, block, shmem, stream, args...) (mcmc.cpp:252, 314, 326, 396, 404,
// 409); here that is a direct call of the native launch point of the same name.
#pragma once
#include "../../sxmc/nll_kernels.h"
#define HEMI_KERNEL_LAUNCH(name, grid, block, shmem, stream, ...) \
  SXMC_KERNEL_LAUNCH(name, grid, block, shmem, stream, __VA_ARGS__)
#define checkCuda(x) ::sxmc::check((int)(x))
