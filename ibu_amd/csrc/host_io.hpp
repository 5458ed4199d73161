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
// num_cpus::get()
size_t host_cores();
}  // namespace ibu
