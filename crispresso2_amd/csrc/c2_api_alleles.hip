#include "c2_ctx.h"
#include "c2_k_alleles.hip"
