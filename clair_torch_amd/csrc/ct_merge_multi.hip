// ct_merge_multi.hip -- second translation unit of ct_merge.hip: the several-batches-per-launch instantiations of
// ct::merge_pivot_kernel (ct::merge_pivot_multi), compiled next to the rest of the file so that the build takes half as long.
#define CT_MERGE_PART 1
#include "ct_merge.hip"
