"""Per-launch durations of the ring Fourier kernels from a rocprofv3 kernel trace (one launch per FFT-size class):
   rocprofv3 --kernel-trace -d gpurun_out/fftprof -o fft -- python3 tools/time_fft.py ; python tools/fft_classes.py gpurun_out/fftprof"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "subdft" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"][:60], int(r["Grid_Size_Y"]) if "Grid_Size_Y" in r else 0, int(r.get("Workgroup_Size_X", 0)),
                         int(r.get("LDS_Block_Size", 0) or 0), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
half = rows[len(rows) // 2:]  # second (timed) call
tot = sum(r[5] for r in half)
for r in half:
    print("%-60s rings %5d  threads %4d  lds %6d  %8.3f ms" % r[1:])
print("total %.2f ms" % tot)
