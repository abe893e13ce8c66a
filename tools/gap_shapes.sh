cd /tmp && export TMPDIR=/tmp
for shp in "1024 1024 1024" "2048 2048 1024" "4096 4096 1024"; do
  d=/tmp/gp_$(echo $shp | tr ' ' _)
  rocprofv3 --kernel-trace -d $d -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/shape_profile.py $shp 14 20 > /dev/null 2>&1
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  echo "== $shp (last call = fast mode)"; python $GRAFT_REPO_ROOT/tools/gap_trace.py $f 6
  echo "-- accurate (the last of the first 20 calls)"; python - $f <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("void oz2::","oz2::"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
nacc=next(i for i,r in enumerate(rows) if "fast_shift" in r["Kernel_Name"])  # the accurate-mode calls come first
per=nacc//20
acc=rows[:nacc][-per:]
prev=None
for r in acc:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:60]:60s} {(e-s)/1e3:8.1f} us gap {((s-prev)/1e3 if prev else 0):6.1f}")
    prev=e
a=rows[:nacc][-3*per:]
print("kernels per accurate call", per, "busy/call", sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in a)/3e3, "us; span/call", (int(a[-1]["End_Timestamp"])-int(a[0]["Start_Timestamp"]))/3e3, "us")
PY
done
