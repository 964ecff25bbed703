# A/B of the packed STFT kernel's pass order (DSA_STFT_RUN = run length; 0: round-robin) on one box
for r in 1 2; do
for R in 0 2 4 7 13; do
echo "R=$R: $(DSA_STFT_RUN=$R python tools/time_fused_mcep.py 2>/dev/null | tail -1 | sed 's/.*mcep alone/mcep alone/')"
done; done
