/* spmv_amd/hip_check.h -- error convention of the boundary: a failed HIP or
 * RCCL call prints the error with its line and ends the process, as the
 * reference's CUDA_CHECK does (reference include/spmv.h:46-53). Only pulled in
 * by translation units that include <hip/hip_runtime.h> first. */
#ifndef SPMV_AMD_HIP_CHECK_H
#define SPMV_AMD_HIP_CHECK_H

#ifdef HIP_INCLUDE_HIP_HIP_RUNTIME_H
#define HIP_CHECK(call)                                                                    \
    do {                                                                                   \
        hipError_t spmv_amd_err_ = (call);                                                 \
        if (spmv_amd_err_ != hipSuccess) {                                                 \
            fprintf(stderr, "HIP error: %s, %s line %d\n", hipGetErrorString(spmv_amd_err_), \
                    __FILE__, __LINE__);                                                   \
            exit(EXIT_FAILURE);                                                            \
        }                                                                                  \
    } while (0)
#endif

#endif
