import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith("{"):
        d=json.loads(line); print("frames/s %.3e  step %.3f ms | mcep %s %.3f ms (%.1f%%) | stft %.3f ms (%.1f%%)"%(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], 100*d["roofline"]["frac"], d["roofline_stft"]["avg_launch_ms"], 100*d["roofline_stft"]["frac"]))
