// third translation unit of the matrix-core convolution: PSF sizes 29..37 (see the end of ics_conv_mfma.hip)
#define ICS_MFMA_PART 2
#include "ics_conv_mfma.hip"
