// c2_k_alleles.hip -- the allele frequency table on the device (placeholder; filled in below)
#pragma once
#include "c2_k_common.h"
