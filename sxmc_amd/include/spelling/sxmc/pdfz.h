// <sxmc/pdfz.h> as the reference's sources include it (mcmc.h:20).
#pragma once
#include "../hemi/array.h"
#include "../../sxmc/pdfz.h"
