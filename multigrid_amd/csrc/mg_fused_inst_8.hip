// Explicit instantiations of the fused pass (split over several files so that they compile in parallel).
#include "mg_fused_kernel.h"

template int launch_fused<9, true, true, true, false, 1, 3, false>(FusedArgs, hipStream_t);
template int launch_fused<9, true, true, false, false, 1, 3, false>(FusedArgs, hipStream_t);
