// second translation unit of the matrix-core convolution: PSF sizes 19..27 (see the end of ics_conv_mfma.hip)
#define ICS_MFMA_PART 1
#include "ics_conv_mfma.hip"
