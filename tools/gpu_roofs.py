import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vcs-h264_amd")]
import vcs_hip
c = vcs_hip.Context(1080, 1920, 16)
for nb in (1 << 28, 1 << 30, 3 << 29):
    print("bytes %5d MiB  copy(read+write) %8.1f GB/s   fill(write) %8.1f GB/s" %
          (nb >> 20, c.measure_copy_bw(nb, 10), c.measure_fill_bw(nb, 10)))
