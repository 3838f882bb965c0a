// Explicit instantiations of the fused pass: the 9-stage passes (3 sweeps from zero + prolongation + 6 sweeps).
#include "mg_fused_kernel.h"

template int launch_fused<9, true, true, true, false, 1, 3, false>(FusedArgs, hipStream_t);
template int launch_fused<9, true, true, false, false, 1, 3, false>(FusedArgs, hipStream_t);
template int launch_fused<9, true, true, true, false, 1, 3, true>(FusedArgs, hipStream_t);
template int launch_fused<9, true, true, false, false, 1, 3, true>(FusedArgs, hipStream_t);
