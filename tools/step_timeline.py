"""timeline of ONE steady-state training step from a rocprofv3 --kernel-trace CSV: start offset / duration / queue of every dispatch between two
consecutive token_block_fwd launches (tools/final_profiles.sh).   python tools/step_timeline.py <kernel_trace.csv> [step index from the end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "token_block_fwd" in r["Kernel_Name"]]
# a step begins with its weight-prep launch (one before the fused forward); take the step `back` from the end
i0 = starts[-back - 1]
i1 = starts[-back]
while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 3000 and "adamw" not in rows[i0 - 1]["Kernel_Name"]:
    i0 -= 1
while i1 > i0 and "adamw" not in rows[i1 - 1]["Kernel_Name"]:
    i1 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
end_prev = 0
print(f"# {i1 - i0} dispatches, step span {(int(rows[i1 - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
print("# start_us  dur_us  gap_after_prev_end_us  queue  kernel")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("eeg::", "")[:52]
    print(f"{s / 1e3:8.1f} {(e - s) / 1e3:7.1f} {(s - end_prev) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>2}  {name}")
    end_prev = max(end_prev, e)
