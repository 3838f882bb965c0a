// Explicit instantiations of the fused pass (one slice of them per unit: they compile in parallel).
#include "mg_fused_kernel.h"

template int pick_mode<1, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
template int pick_mode<2, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
template int pick_mode<3, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
