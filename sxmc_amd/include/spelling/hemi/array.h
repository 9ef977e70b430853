// hemi/array.h -- SPELLING ONLY.  The reference's callers say `hemi::Array<T>`; hemi itself (an
// un-vendored CUDA portability library) is gone.  This header gives that spelling to the native
// HIP-backed sxmc::DeviceArray<T> so that mcmc.cpp-shaped code compiles unchanged.  No CUDA path,
// no host-only mode, no portability macros.
#pragma once
#include "../../sxmc/device_array.h"
namespace hemi {
template <typename T>
using Array = sxmc::DeviceArray<T>;
}
