// k_fused.hip - the whole-trajectory fused level programs of the planning path (fused_level.hpp).
#include "host.hpp"

namespace mpdx {

int launch_fused_args(const mpdx_unet::Fused& f, const FusedArgs& a, int B, hipStream_t st, bool save) {
    if (save) return launch_fused_train(f, a, B, st);
    switch (f.program) {
        case 0:
            if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqDown>)) return rc;
            hipLaunchKernelGGL(fused_program_kernel<FusedSeqDown>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
            break;
        case 1:
            if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqUpA>)) return rc;
            hipLaunchKernelGGL(fused_program_kernel<FusedSeqUpA>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
            break;
        case 2:
            if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqUpB>)) return rc;
            hipLaunchKernelGGL(fused_program_kernel<FusedSeqUpB>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
            break;
        case 3:
            if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqUpAB>)) return rc;
            hipLaunchKernelGGL(fused_program_kernel<FusedSeqUpAB>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
            break;
        case 4:
            if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqMid2>)) return rc;
            hipLaunchKernelGGL(fused_program_kernel<FusedSeqMid2>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
            break;
        case 5:
            if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqDown3>)) return rc;
            hipLaunchKernelGGL(fused_program_kernel<FusedSeqDown3>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
            break;
        case 6:
            if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqMid3>)) return rc;
            hipLaunchKernelGGL(fused_program_kernel<FusedSeqMid3>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
            break;
        default:
            if (int rc = raise_lds_limit((const void*)fused_level_kernel<false>)) return rc;
            hipLaunchKernelGGL(fused_level_kernel<false>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
    }
    return 0;
}

}  // namespace mpdx
