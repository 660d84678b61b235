"""Sample socket power and shader clock (rocm-smi / sysfs) while bench.py runs a long timed region.
usage (GPU box): python tools/power_sample.py [--steps 600]  -> prints idle and loaded power / clock statistics.
Evidence for DESIGN.md section 3.5: is the forward power-bound?"""
import glob, json, os, re, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "600"


def smi():
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        js = json.loads(out)
        card = next(iter(js.values()))
        return {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "temperature (sensor junction", "fclk"))}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def caps():
    try:
        return subprocess.run(["rocm-smi", "-d", "0", "--showmaxpower", "--json"], capture_output=True, text=True,
                              timeout=10).stdout.strip()
    except Exception as e:  # noqa: BLE001
        return str(e)


print("max power:", caps())
idle = [smi() for _ in range(3)]
print("idle:", idle[-1])
p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "20", "--no-cpu-baseline",
                      "--video-frames", "0", "--no-fp32"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
t0 = time.time()
while p.poll() is None:
    s = smi()
    s["t"] = round(time.time() - t0, 2)
    samples.append(s)
    time.sleep(0.25)
line = [l for l in p.stdout.read().splitlines() if l.startswith("{")]
res = json.loads(line[-1]) if line else {}
print("bench:", res.get("value"), "frames/s,", res.get("ms_per_step"), "ms/step over", steps, "steps")


def num(v):
    m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
    return float(m.group(0)) if m else None


keys = sorted({k for s in samples for k in s if k != "t"})
for k in keys:
    vals = [num(s.get(k)) for s in samples if num(s.get(k)) is not None]
    if vals:
        hot = sorted(vals)[len(vals) // 4:]  # upper three quarters: the loaded part of the run
        print(f"{k}: n={len(vals)} max {max(vals):.1f} median-of-loaded {statistics.median(hot):.1f} min {min(vals):.1f}")
print("last samples:", samples[-3:])
