#!/bin/bash
# What the box's tmpfs offers for large folios: the writer pool's cost per byte is page allocation
# (4 KiB at a time), not copying.  Reports the THP settings for shmem and, if this container may remount,
# the write rate of /dev/shm with huge=always.
echo "shmem_enabled: $(cat /sys/kernel/mm/transparent_hugepage/shmem_enabled 2>&1)"
echo "enabled: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>&1)"
grep -E "/dev/shm|tmpfs" /proc/mounts | head -5
uname -r
python tools/exp_tmpfs.py 2>&1 | grep "fresh" | sed -n '1p;4p;6p'
if mount -o remount,huge=always /dev/shm 2>/dev/null; then
  echo "remounted /dev/shm with huge=always"
  grep "/dev/shm" /proc/mounts
  python tools/exp_tmpfs.py 2>&1 | grep "fresh" | sed -n '1p;4p;6p'
  grep -i "ShmemHugePages\|ShmemPmdMapped" /proc/meminfo
  mount -o remount,huge=never /dev/shm
else
  echo "remount refused"
  mkdir -p /tmp/aasr_huge && mount -t tmpfs -o huge=always,size=40g tmpfs /tmp/aasr_huge 2>&1 && echo "own tmpfs mounted" && umount /tmp/aasr_huge
fi
