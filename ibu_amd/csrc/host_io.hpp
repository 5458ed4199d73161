// host_io.hpp — internal hooks the streaming layer (stream.cpp) needs from host_io.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/ibu_hip.h"

namespace ibu {
// Writer::write_slice (writer.rs:321-351) on raw bytes — used by the device write_batch path.
int32_t writer_write_bytes(ibu_writer_t* w, const uint8_t* bytes, size_t len);
// Front half of load_to_vec (reader.rs:511-526): open, header, validate, size check.
int32_t open_plain_file(const char* path, int* fd_out, ibu_header_t* header, size_t* n_records);
// The device streaming path's refill: reads straight from the reader's source into `dst` (a pinned ring slot) until
// `cap_bytes` are there or the stream ends, without the detour through the reader's own 1.18 MB buffer (which must be
// empty: ibu_reader_buffered == 0).  Whole records only: a stream that ends inside a record is TruncatedRecord with the
// position read_batch would report (reader.rs:232-237), with *got_bytes = the complete record bytes in front of the cut and
// r->bytes_read unchanged; a source error likewise leaves in *got_bytes the complete record bytes read in front of it.
// *eof: the stream ended (possibly with *got_bytes > 0).
int32_t reader_read_direct(ibu_reader_t* r, uint8_t* dst, size_t cap_bytes, size_t* got_bytes, bool* eof);
const char* reader_bgzf_path_if_untouched(const ibu_reader_t* r);   // host_io.cpp
void reader_set_drained(ibu_reader_t* r, uint64_t records);
// num_cpus::get()
size_t host_cores();
// Threads for the inflate workers (BGZF blocks, pgzip chunks): the CPUs this process may run on, but at most twice its
// cgroup CPU quota.  Twice, because the batch pipeline has serial stretches (windows, hand-over, the consumer's copies) in
// which the quota goes unused and a decode thread stalls on a table-lookup chain that a second thread on the core's other
// hardware thread fills: measured on the 16-CPU-quota box of this pool, 32 threads inflate 1.37x (1e9 records, sustained,
// throttling included) to 1.47x (1e8) faster than 16.  The quota still bounds the CPU time; without a quota this is the
// affinity count.
size_t inflate_threads();
}  // namespace ibu
