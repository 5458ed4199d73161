// numa.cpp — see numa.hpp.  Host only; nothing here touches HIP.
#include "numa.hpp"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "common.hpp"

namespace ibu {

int parse_cpulist(const char* s, cpu_set_t* set) {
  CPU_ZERO(set);
  int count = 0;
  const char* p = s;
  while (*p) {
    while (*p == ' ' || *p == '\n' || *p == '\t' || *p == ',') ++p;
    if (!*p) break;
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p || a < 0) return -1;
    long b = a;
    p = end;
    if (*p == '-') {
      ++p;
      b = strtol(p, &end, 10);
      if (end == p || b < a) return -1;
      p = end;
    }
    for (long c = a; c <= b; ++c)
      if (c < CPU_SETSIZE && !CPU_ISSET((int)c, set)) { CPU_SET((int)c, set); ++count; }
    if (*p && *p != ',' && *p != '\n' && *p != ' ') return -1;
  }
  return count;
}

static bool read_small_file(const char* path, char* buf, size_t cap) {
  FILE* f = fopen(path, "r");
  if (!f) return false;
  size_t k = fread(buf, 1, cap - 1, f);
  fclose(f);
  while (k && (buf[k - 1] == '\n' || buf[k - 1] == ' ')) --k;
  buf[k] = 0;
  return true;
}

void numa_lookup(const char* sysfs_root, const char* bdf, const cpu_set_t* restrict_to, NumaPlace* out) {
  *out = NumaPlace();
  if (!bdf || !*bdf) return;
  const char* root = sysfs_root && *sysfs_root ? sysfs_root : "/sys";
  char path[512], buf[256], lower[64];
  size_t i = 0;
  for (; bdf[i] && i + 1 < sizeof lower; ++i) lower[i] = (char)((bdf[i] >= 'A' && bdf[i] <= 'F') ? bdf[i] - 'A' + 'a' : bdf[i]);   // sysfs spells bus ids in lower case
  lower[i] = 0;
  snprintf(path, sizeof path, "%s/bus/pci/devices/%s/numa_node", root, lower);
  if (!read_small_file(path, buf, sizeof buf)) return;
  char* end = nullptr;
  const long node = strtol(buf, &end, 10);
  if (end == buf || node < 0) return;              // "-1": the platform does not say
  snprintf(path, sizeof path, "%s/devices/system/node/node%ld/cpulist", root, node);
  if (!read_small_file(path, buf, sizeof buf)) { out->node = (int)node; return; }
  snprintf(out->cpulist, sizeof out->cpulist, "%s", buf);
  out->node = (int)node;
  cpu_set_t all;
  if (parse_cpulist(buf, &all) <= 0) return;
  cpu_set_t mine;
  if (restrict_to) mine = *restrict_to;
  else if (sched_getaffinity(0, sizeof mine, &mine) != 0) return;
  CPU_AND(&out->cpus, &all, &mine);
  out->ncpus = CPU_COUNT(&out->cpus);
}

// set_mempolicy / move_pages by number: libnuma is not part of the image, and the two calls are all this needs
PreferNode::PreferNode(int node) {
  if (node < 0 || node >= 1024) return;
  unsigned long mask[16] = {0};
  mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
  if (syscall(SYS_get_mempolicy, &saved_mode_, saved_mask_, (unsigned long)(sizeof saved_mask_ * 8), nullptr, 0ul) != 0) {
    saved_mode_ = 0;                   // (cannot ask: the default policy comes back afterwards)
    memset(saved_mask_, 0, sizeof saved_mask_);
  }
  active_ = syscall(SYS_set_mempolicy, 1 /*MPOL_PREFERRED*/, mask, (unsigned long)(sizeof mask * 8)) == 0;
}
PreferNode::~PreferNode() {
  if (!active_) return;
  if (saved_mode_ == 0 /*MPOL_DEFAULT*/ ||
      syscall(SYS_set_mempolicy, saved_mode_, saved_mask_, (unsigned long)(sizeof saved_mask_ * 8)) != 0)
    (void)syscall(SYS_set_mempolicy, 0 /*MPOL_DEFAULT*/, nullptr, 0ul);
}

RunOnNode::RunOnNode(const NumaPlace& place) {
  if (place.ncpus <= 0) return;
  if (sched_getaffinity(0, sizeof saved_, &saved_) != 0) return;
  active_ = sched_setaffinity(0, sizeof place.cpus, &place.cpus) == 0;
}
RunOnNode::~RunOnNode() {
  if (active_) (void)sched_setaffinity(0, sizeof saved_, &saved_);
}

int node_of_address(const void* p) {
  const long pg = sysconf(_SC_PAGESIZE);
  void* page = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)(pg - 1));
  int status = -1;
  if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) != 0) return -1;
  return status >= 0 ? status : -1;                // -EFAULT / -ENOENT: not mapped / not touched
}
int node_of_range(const void* p, size_t bytes, int pages) {
  if (!p || bytes == 0 || pages < 1) return -1;
  int votes[64] = {0};
  int best = -1;
  for (int i = 0; i < pages; ++i) {
    const int nd = node_of_address(static_cast<const uint8_t*>(p) + (bytes - 1) / (size_t)pages * (size_t)i);
    if (nd >= 0 && nd < 64 && ++votes[nd] > (best < 0 ? 0 : votes[best])) best = nd;
  }
  return best;
}

}  // namespace ibu

// Where a PCI function hangs off the host: *node = its NUMA node (-1: unknown), cpulist = that node's CPUs as sysfs spells them
// ("" when unknown), *usable_cpus (nullable) = how many of them the calling thread may run on.
extern "C" int32_t ibu_numa_of_pci(const char* sysfs_root, const char* pci_bus_id, int32_t* node, char* cpulist, size_t cap,
                                   int32_t* usable_cpus) {
  if (!pci_bus_id || !node) return ibu::err_arg("pci_bus_id or node is NULL");
  ibu::NumaPlace pl;
  ibu::numa_lookup(sysfs_root, pci_bus_id, nullptr, &pl);
  *node = pl.node;
  if (cpulist && cap) snprintf(cpulist, cap, "%s", pl.cpulist);
  if (usable_cpus) *usable_cpus = pl.ncpus;
  return IBU_OK;
}
