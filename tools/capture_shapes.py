"""Which multi-stream fork/join shapes this ROCm's HIP-graph stream capture accepts (torch streams/events, one child process per
shape). Round 3 finding on ROCm 7.2 / MI355X: a stream forked from a forked stream is fine as long as every forked stream joins
the ORIGIN stream directly; joining a grandchild stream back into its parent stream (s3 -> s2 -> s1) segfaults inside the capture.
The frame driver's description stream therefore joins the caller's stream, not the side stream (csrc/nm_frame.hip)."""
import subprocess, sys
SRC = '''
import torch, sys
mode = sys.argv[1]
dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev = [torch.cuda.Event() for _ in range(8)]
def body():
    x.add_(1)
    ev[0].record(s1); s2.wait_event(ev[0])
    with torch.cuda.stream(s2):
        y = x * 2
        ev[1].record(s2)
    if mode == "two":
        s1.wait_event(ev[1]); return y, y
    if mode in ("siblings", "cross"):
        ev[2].record(s1); s3.wait_event(ev[2])
    if mode in ("cross", "nested", "nested_join_parent"):
        s3.wait_event(ev[1])
    with torch.cuda.stream(s3):
        z = (y if mode != "siblings" else x) + 1
        ev[3].record(s3)
    if mode == "nested_join_parent":          # s3 joins s2 (its parent), s2 joins s1: crashes in capture on ROCm 7.2
        with torch.cuda.stream(s2):
            s2.wait_event(ev[3]); ev[4].record(s2)
        s1.wait_event(ev[4])
        return z, y
    s1.wait_event(ev[1]); s1.wait_event(ev[3])
    return z, y
with torch.cuda.stream(s1):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s1):
    z, w = body()
g.replay(); torch.cuda.synchronize()
print(mode, "capture ok", float(z[0]), float(w[0]))
'''
# The known-crashing shape (a process that holds the GPU segfaults inside the capture) only runs when asked for:
#   python tools/capture_shapes.py --include-crashing
modes = ["two", "siblings", "cross", "nested"] + (["nested_join_parent"] if "--include-crashing" in sys.argv[1:] else [])
for m in modes:
    r = subprocess.run([sys.executable, "-c", SRC, m], capture_output=True, text=True)
    print(m, "rc", r.returncode, r.stdout.strip()[-80:], r.stderr.strip()[-120:].replace("\n", " | "))
