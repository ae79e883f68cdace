"""Shader clock and socket power while bench.py runs its timed steps (the counter passes drain between dispatches and report the clock of
a cooled-down chip; this samples the running step):  python3 tools/clock_sampler.py [bench.py flags ...]
Starts bench.py as a child with --steps 150, polls `rocm-smi --showclocks --showpower` every 0.25 s, prints the samples taken while the GPU
was busy."""
import re, subprocess, sys, time
child = subprocess.Popen([sys.executable, "bench.py", "--steps", "150", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs", "--no-kernel-events"] + sys.argv[1:],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
while child.poll() is None:
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
    except Exception as e:
        out = ""
    sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    pw = re.findall(r"Power \(W\): ([\d.]+)", out)
    if sclk or pw:
        samples.append((time.time(), [int(x) for x in sclk], [float(x) for x in pw]))
    time.sleep(0.25)
line = child.stdout.read().strip().splitlines()[-1] if child.stdout else ""
print("bench:", line[:160])
busy = [s for s in samples if s[2] and max(s[2]) > 400]
print(f"{len(samples)} samples, {len(busy)} with socket power > 400 W")
for t, c, p in busy[:: max(1, len(busy) // 24)]:
    print(f"  sclk {c} MHz   power {p} W")
if busy:
    print("mean sclk of busy samples:", sum(sum(c) / max(len(c), 1) for _, c, _ in busy) / len(busy), "MHz; mean power", sum(max(p) for _, _, p in busy) / len(busy), "W")
