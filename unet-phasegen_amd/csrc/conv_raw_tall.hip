// conv_raw_tall.hip -- raw-window F / T kernels, workgroup tile 256 (M) x 128 (N): problems with few columns (inference).
#include "conv_raw_impl.h"

hipError_t pgconv::launch_raw_ft_tall(int kind, const IgemmParams& p, int grid, hipStream_t st, int prec) {
    return launch_raw_ft_wn<1>(kind, p, grid, st, prec);
}
