// sampler_grp_all: the sixteen-lane paired-end bodies of every isoform-count class (K <= 4, 8, 12, 16, 32) behind one entry point
// (see kernels_grp.inl)
#define MISO_GRP_ALL_CLASSES 1
#include "kernels_grp.inl"
