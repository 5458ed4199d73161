/* Plain C11 client of the C ABI (what a cgo / Rust extern "C" / JNI stub sees): no C++, no torch, just ibu_hip.h.
 * Writes the README's two-record file into memory, reads it back, checks sizes, bytes and one error payload. */
#include <stdio.h>
#include <string.h>

#include "ibu_hip.h"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "abi_smoke: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
  REQUIRE(sizeof(ibu_header_t) == IBU_HEADER_SIZE && sizeof(ibu_record_t) == IBU_RECORD_SIZE);
  ibu_header_t h;
  ibu_header_init(&h, 16, 12);
  ibu_header_set_sorted(&h);
  REQUIRE(ibu_header_validate(&h) == IBU_OK && ibu_header_sorted(&h) == 1);

  ibu_writer_t* w = NULL;
  REQUIRE(ibu_writer_open_mem(&h, &w) == IBU_OK);
  const ibu_record_t recs[2] = {{0x1100, 0x100011, 0}, {0x1101, 0x100010, 1}};
  REQUIRE(ibu_writer_write_batch(w, recs, 2) == IBU_OK && ibu_writer_finish(w) == IBU_OK);
  REQUIRE(ibu_writer_records_written(w) == 2);
  uint8_t* bytes = NULL;
  size_t len = 0;
  REQUIRE(ibu_writer_into_inner(w, &bytes, &len) == IBU_OK && len == 80);  /* README.md:84-85 */
  REQUIRE(memcmp(bytes, "IBU!", 4) == 0 && bytes[16] == 1);

  ibu_reader_t* r = NULL;
  REQUIRE(ibu_reader_open_mem(bytes, len, &r) == IBU_OK);
  ibu_record_t got;
  int32_t have = 0;
  REQUIRE(ibu_reader_next(r, &got, &have) == IBU_OK && have == 1 && got.barcode == 0x1100 && got.umi == 0x100011);
  REQUIRE(ibu_reader_next(r, &got, &have) == IBU_OK && have == 1 && ibu_record_cmp(&recs[1], &got) == 0);
  REQUIRE(ibu_reader_next(r, &got, &have) == IBU_OK && have == 0);
  ibu_reader_close(r);
  ibu_free(bytes);

  h.magic = 0x12345678u; /* error payloads travel through ibu_last_error, not through exceptions */
  REQUIRE(ibu_header_validate(&h) == IBU_ERR_INVALID_MAGIC);
  ibu_error_detail_t d;
  ibu_last_error(&d);
  REQUIRE(d.code == IBU_ERR_INVALID_MAGIC && d.a == IBU_MAGIC && d.b == 0x12345678u && strstr(d.message, "0x12345678"));
  REQUIRE(strcmp(ibu_status_name(IBU_ERR_TRUNCATED_RECORD), "TruncatedRecord") == 0 && ibu_abi_revision() >= 1);

  size_t s = 0, e = 0;
  REQUIRE(ibu_shard_range(10, 3, 2, &s, &e) == IBU_OK && s == 6 && e == 10); /* mmap.rs:297-307 */
  ibu_ctx_t* ctx = NULL;
  int32_t ndev = 0;
  if (ibu_device_count(&ndev) != IBU_OK || ndev == 0)
    REQUIRE(ibu_ctx_create(0, &ctx) == IBU_ERR_NO_DEVICE && ctx == NULL); /* no CPU fallback behind the device API */
  printf("abi_smoke ok (%s)\n", ibu_version());
  return 0;
}
