"""Print the launch sequence (short kernel name, grid, duration us) of a window of a rocprofv3 kernel trace."""
import csv, sys
f, start, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = {"tgemm_kernel<float, float, float, float": "gemm", "tgemm_kernel<float, float, double, double": "GRAM",
         "chol_upper": "CHOL", "jacobi_rows_reg256": "JREG", "jacobi_rows_kernel": "JGEN", "select_rows": "sel",
         "normalize": "norm", "fill_kernel": "fill", "copyBuffer": "copy", "fillBuffer": "memset", "add_logs": "addl"}
for r in rows[start:start + count]:
    n = r["Kernel_Name"]
    s = next((v for k, v in short.items() if k in n), n[:20])
    print("%-6s grid=(%s,%s,%s) %8.1f us" % (s, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"],
                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
