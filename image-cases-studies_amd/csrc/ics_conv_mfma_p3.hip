// fourth translation unit of the matrix-core convolution: PSF sizes 39..49 (see the end of ics_conv_mfma.hip)
#define ICS_MFMA_PART 3
#include "ics_conv_mfma.hip"
