// k_fused_train.hip - the fused level programs in their training-forward variant (every op also stores its output and its GroupNorm
// input, FusedArgs::save): the programs of the two standard networks (four levels: Down3 + UpAB; three levels: Down + Mid3 + UpAB) and the generic op-list kernel.
#include "host.hpp"

namespace mpdx {

int launch_fused_train(const mpdx_unet::Fused& f, const FusedArgs& a, int B, hipStream_t st) {
    if (!fused_save_variant(f)) return fail(MPDX_E_STATE, "fused program %d has no training variant", f.program);
    if (f.program == 3) {
        if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqUpAB, true>)) return rc;
        hipLaunchKernelGGL((fused_program_kernel<FusedSeqUpAB, true>), dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
    } else if (f.program == 5) {
        if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqDown3, true>)) return rc;
        hipLaunchKernelGGL((fused_program_kernel<FusedSeqDown3, true>), dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
    } else if (f.program == 0) {   // the three-level network (round 6): downs.0 + downs.1 ...
        if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqDown, true>)) return rc;
        hipLaunchKernelGGL((fused_program_kernel<FusedSeqDown, true>), dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
    } else if (f.program == 6) {   // ... and downs.2 + the middle blocks
        if (int rc = raise_lds_limit((const void*)fused_program_kernel<FusedSeqMid3, true>)) return rc;
        hipLaunchKernelGGL((fused_program_kernel<FusedSeqMid3, true>), dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
    } else {
        if (int rc = raise_lds_limit((const void*)fused_level_kernel<true>)) return rc;
        hipLaunchKernelGGL(fused_level_kernel<true>, dim3(B), dim3(kFusedThreads), f.lds_bytes, st, a);
    }
    return 0;
}

}  // namespace mpdx
