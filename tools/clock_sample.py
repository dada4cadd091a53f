"""Sample the GPU's shader clock and power from a SEPARATE process while a workload runs (the claim that the headline
kernel is slower than its own loads and stores because the chip clocks down under fp64 MFMA load rested on GRBM_GUI_ACTIVE / time).
    python tools/clock_sample.py out.csv [hz] &      ... run the workload ...      kill $!
Sources, in this order: the amdgpu hwmon files of the first card that has them (freq1_input = sclk in Hz, power1_average / power1_input
in microwatts; no tool start-up per sample, tens of samples per second), else `rocm-smi --showclocks --showpower --json` in a loop.
This process never initialises HIP."""
import glob
import json
import os
import signal
import subprocess
import sys
import time

out = open(sys.argv[1], "w")
hz = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
run = [True]
signal.signal(signal.SIGTERM, lambda *_: run.__setitem__(0, False))
signal.signal(signal.SIGINT, lambda *_: run.__setitem__(0, False))


def rd(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None


hws = [d for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if rd(os.path.join(d, "freq1_input")) is not None]
if hws:
    # a GPU box shows the hwmon files of every card of its host; the card under test is the one whose power moves (the summary picks it)
    pws = ["power1_average" if rd(os.path.join(d, "power1_average")) is not None else "power1_input" for d in hws]
    out.write(f"# source: amdgpu hwmon (freq1_input Hz = sclk, power uW) of {len(hws)} cards, target {hz} Hz\n")
    out.write("# t_s," + ",".join(f"{d.split('/')[4]}_sclk_MHz,{d.split('/')[4]}_power_W" for d in hws) + "\n")
    t0 = time.perf_counter()
    while run[0]:
        cols = []
        for d, pw in zip(hws, pws):
            f, p = rd(os.path.join(d, "freq1_input")), rd(os.path.join(d, pw))
            cols += [str(int(f) / 1e6) if f else "", str(int(p) / 1e6) if p else ""]
        out.write(f"{time.perf_counter() - t0:.3f}," + ",".join(cols) + "\n")
        out.flush()
        time.sleep(1.0 / hz)
else:
    out.write("# source: rocm-smi --showclocks --showpower --json (one tool start per sample)\n# t_s,sclk,power\n")
    t0 = time.perf_counter()
    while run[0]:
        try:
            j = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout)
            c = j.get("card0", {})
            sclk = next((v for k, v in c.items() if "sclk" in k.lower()), "")
            power = next((v for k, v in c.items() if "power" in k.lower()), "")
            out.write(f"{time.perf_counter() - t0:.3f},{sclk},{power}\n")
        except Exception as e:  # keep sampling
            out.write(f"{time.perf_counter() - t0:.3f},error,{type(e).__name__}\n")
        out.flush()
out.close()
