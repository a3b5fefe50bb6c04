"""Socket power and shader clock of the GPU while the bench runs: a sidecar that never touches the GPU samples the amdgpu
hwmon files (power1_input [uW], power1_cap, freq1_input = sclk [Hz]; readable by an ordinary user on the GPU box) at ~50 Hz
from a thread, and runs each phase as a child process:

  idle             nothing running
  bench_in_situ    bench.py, default configuration: the product on its side stream beside the rest of the step
  bench_serial     MCGRA_OVERLAP=0: the product alone on the chip, the rest of the step behind it
  product_alone    scripts/product_loop.py: packs + split2_m16_kernel back to back

The box exposes every GPU of the node in sysfs but only one to the process: the active card is the one whose power moves.
Output: JSON with per-phase statistics of the busy window and the raw samples of the active card.

    python scripts/power_trace.py --out profiles/r04_power_trace.json [--steps 600]
"""
import argparse, glob, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_int(path):
    try:
        with open(path) as fh:
            return int(fh.read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, cards, hz):
        super().__init__(daemon=True)
        self.cards, self.dt, self.rows, self.stop_flag, self.phase = cards, 1.0 / hz, [], False, "idle"

    def run(self):
        while not self.stop_flag:
            t = time.time()
            row = [t, self.phase]
            for c in self.cards:
                row.append((read_int(c + "/power1_input"), read_int(c + "/freq1_input")))
            self.rows.append(row)
            time.sleep(max(0.0, self.dt - (time.time() - t)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--hz", type=float, default=50.0)
    a = ap.parse_args()
    cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    cards = [c for c in cards if os.path.exists(c + "/power1_input")]
    if not cards:
        raise SystemExit("no readable amdgpu hwmon power1_input")
    caps = [read_int(c + "/power1_cap") for c in cards]
    s = Sampler(cards, a.hz)
    s.start()
    py = sys.executable
    bench = [py, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "5", "--no-cpu-baseline", "--no-split-probe"]
    phases = [("idle", None, {}),
              ("bench_in_situ", bench, {}),
              ("cooldown1", None, {}),
              ("bench_serial", bench, {"MCGRA_OVERLAP": "0", "MCGRA_AB": "1"}),
              ("cooldown2", None, {}),
              ("product_alone", [py, os.path.join(ROOT, "scripts", "product_loop.py"), "--seconds", "5"], {})]
    lines = {}
    for name, cmd, env in phases:
        s.phase = name
        if cmd is None:
            time.sleep(2.0)
            continue
        r = subprocess.run(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        for ln in r.stdout.splitlines():
            if ln.startswith("{"):
                try:
                    lines[name] = json.loads(ln)
                except ValueError:
                    pass
    s.phase = "end"
    time.sleep(0.5)
    s.stop_flag = True
    s.join()
    rows = s.rows
    # the active card: largest power swing
    swing = []
    for k in range(len(cards)):
        p = [r[2 + k][0] for r in rows if r[2 + k][0] is not None]
        swing.append((max(p) - min(p)) if p else 0)
    k = max(range(len(cards)), key=lambda i: swing[i])
    idle = [r[2 + k][0] for r in rows if r[1] == "idle" and r[2 + k][0] is not None]
    idle_w = sum(idle) / max(1, len(idle)) / 1e6
    out = {"what": __doc__.split("\n\n")[0], "card": cards[k], "power_cap_w": (caps[k] or 0) / 1e6, "idle_w": idle_w,
           "sample_hz": a.hz, "other_cards_swing_w": [x / 1e6 for i, x in enumerate(swing) if i != k], "phases": {}}
    for name, cmd, env in phases:
        if cmd is None:
            continue
        smp = [(r[0], r[2 + k][0] / 1e6, (r[2 + k][1] or 0) / 1e6) for r in rows if r[1] == name and r[2 + k][0] is not None]
        if not smp:
            continue
        pmax = max(x[1] for x in smp)
        busy = [x for x in smp if x[1] >= idle_w + 0.6 * (pmax - idle_w)]
        pw = sorted(x[1] for x in busy)
        ck = sorted(x[2] for x in busy)
        ph = {"samples": len(smp), "busy_samples": len(busy), "power_w_mean": sum(pw) / len(pw), "power_w_median": pw[len(pw) // 2],
              "power_w_max": pw[-1], "sclk_mhz_mean": sum(ck) / len(ck), "sclk_mhz_median": ck[len(ck) // 2], "sclk_mhz_min": ck[0],
              "sclk_mhz_max": ck[-1], "frac_of_cap_mean": sum(pw) / len(pw) / max(1e-9, (caps[k] or 0) / 1e6), "env": env}
        ln = lines.get(name)
        if ln:
            ph["result"] = {kk: ln[kk] for kk in ("value", "ms_per_step", "ms_per_call", "calls") if kk in ln}
            if "roofline" in ln and ln["roofline"]:
                ph["result"]["product_avg_launch_ms"] = ln["roofline"].get("avg_launch_ms")
        ph["trace"] = [[round(x[0] - rows[0][0], 3), round(x[1], 1), round(x[2], 0)] for x in smp]
        out["phases"][name] = ph
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(out, fh)
    print(json.dumps({kk: {x: v[x] for x in v if x != "trace"} for kk, v in out["phases"].items()}, indent=1))
    print("card", out["card"], "cap", out["power_cap_w"], "idle", out["idle_w"])


if __name__ == "__main__":
    main()
