// numa.hpp — where a device hangs off the host, and keeping its feed on that side (no libnuma: sysfs + three syscalls).
//
// The reference's process_parallel gives every worker its range of the map and lets the kernel place the pages
// (mmap.rs:297-322).  With a GPU per worker the range crosses PCIe, and on a two-socket node the copy page cache -> pinned
// ring -> device is only local if the ring and the threads that fill it sit on the socket the GPU hangs off: device ->
// PCI bus id -> /sys/bus/pci/devices/<bdf>/numa_node -> /sys/devices/system/node/node<N>/cpulist.
#pragma once
#include <sched.h>
#include <stddef.h>
#include <stdint.h>

namespace ibu {

struct NumaPlace {
  int node = -1;        // NUMA node of the device's PCI function; -1: unknown (no sysfs, a VM, a single-node host that says -1) or option "numa" = 0
  int ncpus = 0;        // CPUs of that node this process may run on (0: nothing to pin to)
  cpu_set_t cpus;       // valid when ncpus > 0
  char cpulist[256];    // the node's cpulist as sysfs spells it (whole node, before the intersection with the process's mask)
  NumaPlace() { CPU_ZERO(&cpus); cpulist[0] = 0; }
};

// "0-15,128-143" -> set; returns the number of CPUs, -1 on a malformed list.
int parse_cpulist(const char* s, cpu_set_t* set);
// sysfs_root: "/sys" (NULL) or a test tree.  bdf: "0000:c1:00.0".  Fills *out (node -1 when the tree says so or lacks the file).
// `restrict_to`: the mask to intersect the node's CPUs with (NULL: this thread's current affinity).
void numa_lookup(const char* sysfs_root, const char* bdf, const cpu_set_t* restrict_to, NumaPlace* out);

// For the lifetime of the object the calling thread prefers `node` for new pages (set_mempolicy MPOL_PREFERRED); active() says
// whether the kernel accepted it (a container's seccomp profile may not: then allocations go where the runtime puts them).
class PreferNode {
 public:
  explicit PreferNode(int node);
  ~PreferNode();
  bool active() const { return active_; }
 private:
  bool active_ = false;
  int saved_mode_ = 0;                 // the thread's policy before (a process under `numactl --interleave` gets it back)
  unsigned long saved_mask_[16] = {0};
};
// For the lifetime of the object the calling thread (and every thread it starts: affinity is inherited) runs on place.cpus.
class RunOnNode {
 public:
  explicit RunOnNode(const NumaPlace& place);
  ~RunOnNode();
  bool active() const { return active_; }
 private:
  cpu_set_t saved_;
  bool active_ = false;
};
// Node that holds the page at `p` (move_pages in query mode; the page must have been touched), or -1 when the kernel will not say.
int node_of_address(const void* p);
// The node most of `pages` sampled pages of [p, p + bytes) are on; -1 unknown.
int node_of_range(const void* p, size_t bytes, int pages = 16);

}  // namespace ibu
