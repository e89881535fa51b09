"""Does the expansion kernel's time follow the clocks?  Fixed-sweep float32 ticks on smallCapture 16 384 in segments of 125 launches
(the roofline pass of bench.py), with the card's shader / memory clock and power read from sysfs by a sampler thread.
usage (GPU box): python tools/clock_probe.py [segments] -> one line per segment + a summary."""
import glob, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmx


def _cards():
    return sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))


def _active(path):
    try:
        for line in open(path).read().splitlines():
            if line.rstrip().endswith("*"):
                return int("".join(c for c in line.split(":")[1] if c.isdigit()))
    except Exception:
        return None
    return None


def _power(dev_dir):
    for f in glob.glob(dev_dir + "/hwmon/hwmon*/power1_average") + glob.glob(dev_dir + "/hwmon/hwmon*/power1_input"):
        try:
            return int(open(f).read()) / 1e6
        except Exception:
            pass
    return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False
        c = _cards()
        self.dir = os.path.dirname(c[0]) if c else None

    def run(self):
        while not self.stop and self.dir:
            self.rows.append((time.perf_counter(), _active(self.dir + "/pp_dpm_sclk"), _active(self.dir + "/pp_dpm_mclk"), _power(self.dir)))
            time.sleep(0.02)


if __name__ == "__main__":
    n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    env = pmx.PmxVecEnv(pmx.get_layout("smallCapture"), 16384, length=300, auto_reset=True, obs_dtype="float32", device=dev)
    env.reset()
    env.set_tuning("expand_alt", 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    actions = torch.randint(0, 5, (64, 16384, 4), generator=g, device=dev, dtype=torch.int8)
    s = Sampler(); s.start()
    time.sleep(0.3)
    out = []
    for i in range(n_seg):
        t0 = time.perf_counter()
        env.profile_begin(125 + 8)
        for k in range(125):
            env.step(actions[k % 64])
        pr = env.profile_end()
        t1 = time.perf_counter()
        rows = [r for r in s.rows if t0 <= r[0] <= t1] or s.rows[-1:]
        def mean(j):
            v = [r[j] for r in rows if r[j] is not None]
            return round(sum(v) / len(v), 1) if v else None
        rec = {"segment": i, "expand_us": round(pr["expand_ms"] * 1e3 / pr["expand_launches"], 2),
               "rule_us": round(pr["rule_ms"] * 1e3 / max(pr["rule_launches"], 1), 2), "sclk_mhz": mean(1), "mclk_mhz": mean(2), "power_w": mean(3)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
        if i == n_seg // 2:                                   # a pause in the middle: does an idle second bring the first segments' time back?
            torch.cuda.synchronize(dev); time.sleep(3.0)
    s.stop = True
    us = [r["expand_us"] for r in out]
    print(json.dumps({"summary": True, "first3_us": us[:3], "last3_us": us[-3:], "min_us": min(us), "max_us": max(us), "sysfs": s.dir}))
