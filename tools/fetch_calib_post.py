import csv, glob, collections, sys
for tag in ("calib_f", "calib_w"):
    for fn in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            print(tag, r["Kernel_Name"][:40], r["Counter_Name"], float(r["Counter_Value"]))
