// Explicit instantiations of the fused pass (split over several files so that they compile in parallel).
#include "mg_fused_kernel.h"

template int pick_mode<1, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
template int pick_mode<2, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
