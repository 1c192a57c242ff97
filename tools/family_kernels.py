"""Developer tool: the library kernels of one eager training step of a model family, by time (library event timer):
    python tools/family_kernels.py pix2pix256_b64 [top]"""
import json
import subprocess
import sys

fam = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
out = subprocess.run([sys.executable, "bench.py", "--family", fam, "--steps", "5", "--warmup", "2"], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print(fam, "%.3f ms per step" % d["ms_per_step"], "step TFLOP/s %.0f" % d["roofline"]["step_tflops"])
ak = d["roofline"]["all_kernels"]
tot = sum(v["launches"] * v["avg_us"] for v in ak.values())
print("library kernel time of one eager step: %.3f ms" % (tot / 1e3))
for k, v in sorted(ak.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_us"])[:top]:
    print("%6.2f%%  n=%4d  avg=%8.1f us  %7.0f TF  %s" % (100 * v["launches"] * v["avg_us"] / tot, v["launches"], v["avg_us"], v["tflops"], k))
