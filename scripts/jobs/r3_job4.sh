mkdir -p gpurun_out/r3
for spec in "2 5" "2 133" "2 4" "4 133"; do set -- $spec; echo "== parity cpl $1 form $2"; GNUAIS_FIR_CPL=$1 GNUAIS_FIR_FORM=$2 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2; done > gpurun_out/r3/parity_wide2.txt 2>&1
cat gpurun_out/r3/parity_wide2.txt
timeout 900 python scripts/time_fir_wide.py all > gpurun_out/r3/time_fir_wide2.txt 2>&1
cat gpurun_out/r3/time_fir_wide2.txt
python - <<'PY' > gpurun_out/r3/time_fir_elim2.txt 2>&1
import sys; sys.path.insert(0, "scripts"); sys.argv = ["x", "fir"]
exec(open("scripts/time_fir_wide.py").read().split("which = sys.argv")[0])
names = {0: "full kernel", 1: "no exact path", 2: "no sign stores", 4: "no central sum", 8: "no peak", 16: "no epilogue",
         5: "no exact path, no central sum", 27: "only loads + central sum", 31: "only loads"}
for dbg in (0, 1, 2, 4, 8, 16, 5, 27, 31):
    ms = measure(2, 5, 512, 1, extra=dict(fir_dbg=dbg))
    print(f"cpl 2 raw pf2 pk G16 dbg {dbg:2d} ({names[dbg]:32s}): {ms:.3f} ms  {n_ch*total*2/ms/1e9:.2f} TB/s", flush=True)
PY
cat gpurun_out/r3/time_fir_elim2.txt
