# average socket power and shader clock while a kernel loops (run via gpurun): evidence for the power-limit reading of DESIGN 3.2
# usage: tools/power_probe.sh  -> gpurun_out/power_probe.txt
mkdir -p gpurun_out
python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
with torch.no_grad():
    X = stft(x)
    for phase, fn in (("mcep", lambda: mcep(X)), ("stft", lambda: stft(x))):
        open("gpurun_out/power_phase.txt", "w").write(phase)
        t0 = time.time()
        while time.time() - t0 < 6.0:
            for _ in range(200): fn()
            torch.cuda.synchronize()
open("gpurun_out/power_phase.txt", "w").write("idle")
time.sleep(3)
open("gpurun_out/power_phase.txt", "w").write("done")
PY
sleep 4
: > gpurun_out/power_probe.txt
while true; do
  ph=$(cat gpurun_out/power_phase.txt 2>/dev/null)
  [ "$ph" = "done" ] && break
  p=$(rocm-smi --showpower 2>/dev/null | grep "Package Power" | head -1 | sed 's/.*: *//')
  c=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1 | sed 's/.*(//; s/).*//')
  echo "$ph power=$p sclk=$c" >> gpurun_out/power_probe.txt
  sleep 0.5
done
wait
python - <<'PY'
import collections, re
acc = collections.defaultdict(list)
for l in open("gpurun_out/power_probe.txt"):
    m = re.match(r"(\w+) power=([\d.]+) sclk=(\d+)", l)
    if m: acc[m.group(1)].append((float(m.group(2)), int(m.group(3))))
for ph, v in acc.items():
    print(f"{ph}: {len(v)} samples, socket power {min(p for p, _ in v):.0f} .. {max(p for p, _ in v):.0f} W (cap 1400 W), sclk level {min(c for _, c in v)} .. {max(c for _, c in v)} MHz")
PY
