#!/usr/bin/env python3
"""Durations (us) of the launches of kernels whose name contains PATTERN in a rocprofv3 kernel-trace CSV:
    python tools/kdur.py TRACE.csv PATTERN"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}  vgpr {r["VGPR_Count"]} lds {r["LDS_Block_Size"]} scratch {r["Scratch_Size"]}')
