set -e
g++ -O2 -pthread tools/native/filewrite_bench.cpp -o /tmp/fwb
df -hT /tmp /dev/shm | cat
nproc; cat /sys/fs/cgroup/cpu.max
for d in /tmp /dev/shm; do
  for m in "0 1" "1 8" "1 16" "2 8" "2 16" "3 8" "3 16"; do echo -n "$d "; /tmp/fwb $d/fwb.bin $m 12e9; done
done
