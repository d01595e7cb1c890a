// Compile-only probe for tests/test_hot_loop_isa.py: the two headline instantiations of the single-pass kernel,
// exactly as fos_plan.hip's menu instantiates them (cfg2: 512 threads x 4 chunks, 3 register tiles; cfg4: 1024 x 4, drained).
#include "../fastoptsolver_amd/csrc/gemv_pair.hpp"
template __global__ void fos::gemv_pair_kernel<float, 512, 4, 1, true, 2, true, 3, false, false, false>(
    const float*, int64_t, const float*, int64_t, int, fos::YSource, int64_t, float*, double*, double*);
template __global__ void fos::gemv_pair_kernel<float, 1024, 4, 1, true, 4, true, 2, false, false, true>(
    const float*, int64_t, const float*, int64_t, int, fos::YSource, int64_t, float*, double*, double*);
template __global__ void fos::gemv_pair_kernel<fos::bf16_t, 512, 4, 1, true, 2, true, 3, false, false, false>(
    const fos::bf16_t*, int64_t, const float*, int64_t, int, fos::YSource, int64_t, float*, double*, double*);
