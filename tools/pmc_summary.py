"""Average PMC counter values per kernel from a rocprofv3 --pmc ... --output-format csv run.
usage: pmc_summary.py <counter_collection.csv> [min_duration_us per kernel-substring, e.g. nl_find=50]"""
import sys
import pandas as pd

df = pd.read_csv(sys.argv[1])
df["dur"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
mins = dict(a.split("=") for a in sys.argv[2:])
piv = df.pivot_table(index=["Dispatch_Id", "Kernel_Name", "dur", "VGPR_Count", "Accum_VGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size"],
                     columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 40); pd.set_option("display.float_format", lambda v: "%.1f" % v)
names = ["nl_find", "nl_prepare", "force_front", "pairs_fft_plane", "pairs_fft_lines", "nb_direct", "pme_spread", "fft_plane", "fft_kernel", "pme_interp", "k_terms", "k_step_units", "k_clear2"]
rows = []
for name in names:
    sub = piv[piv.Kernel_Name.str.contains(name)]
    if name in mins:
        sub = sub[sub.dur > float(mins[name])]
    if len(sub) == 0:
        continue
    m = sub.drop(columns=["Dispatch_Id", "Kernel_Name"]).mean()
    m["n"] = len(sub)
    m.name = name
    rows.append(m)
print(pd.DataFrame(rows).T.to_string())
