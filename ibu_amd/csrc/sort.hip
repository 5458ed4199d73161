// sort.hip — device sort by (barcode, umi, index).  Placeholder until the radix sort lands:
// reports hipErrorNotSupported so callers fail loudly instead of getting unsorted data.
#include "kernels.h"

namespace ibu {
size_t sort_scratch_bytes(const LaunchCfg&, size_t) { return 16; }
hipError_t launch_sort_records(const LaunchCfg&, void*, void*, size_t, void*, size_t, hipStream_t) {
  return hipErrorNotSupported;
}
}  // namespace ibu
