import os, time, subprocess, sys
def run(cmd):
    t=time.perf_counter(); subprocess.run(cmd, shell=True, stderr=subprocess.DEVNULL); return time.perf_counter()-t
for label, cmd in [("buffered 5.4GB /tmp", "dd if=/dev/zero of=/tmp/ddx bs=64M count=80"),
                   ("direct 5.4GB /tmp", "dd if=/dev/zero of=/tmp/ddx bs=64M count=80 oflag=direct"),
                   ("4 parallel buffered /tmp", "for i in 1 2 3 4; do dd if=/dev/zero of=/tmp/ddx$i bs=64M count=20 & done; wait"),
                   ("buffered 5.4GB /dev/shm", "dd if=/dev/zero of=/dev/shm/ddx bs=64M count=80"),
                   ("4 parallel /dev/shm", "for i in 1 2 3 4; do dd if=/dev/zero of=/dev/shm/ddx$i bs=64M count=20 & done; wait")]:
    subprocess.run("rm -f /tmp/ddx* /dev/shm/ddx*", shell=True)
    dt = run(cmd)
    print(f"{label}: {dt:.2f} s = {5.37/dt:.1f} GB/s", flush=True)
subprocess.run("rm -f /tmp/ddx* /dev/shm/ddx*", shell=True)
