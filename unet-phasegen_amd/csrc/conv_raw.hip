// conv_raw.hip -- raw-window F / T kernels, workgroup tile 128 (M) x 256 (N): the training shape.
#include "conv_raw_impl.h"

hipError_t pgconv::launch_raw_ft(int kind, const IgemmParams& p, int grid, hipStream_t st, int prec) {
    return launch_raw_ft_wn<2>(kind, p, grid, st, prec);
}
