cd /tmp && export TMPDIR=/tmp
for shp in "1024 1024 1024" "2048 2048 1024" "4096 4096 1024"; do
  d=/tmp/gp_$(echo $shp | tr ' ' _)
  rocprofv3 --kernel-trace -d $d -o t --output-format csv -- python /root/repo/tools/shape_profile.py $shp 14 20 > /dev/null 2>&1
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  echo "== $shp (last call = fast mode)"; python /root/repo/tools/gap_trace.py $f 6
  echo "-- accurate (first 20 calls x 10 kernels)"; python - $f <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("void oz2::","oz2::"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
acc=rows[:200][-10:]
prev=None
for r in acc:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:60]:60s} {(e-s)/1e3:8.1f} us gap {((s-prev)/1e3 if prev else 0):6.1f}")
    prev=e
a=rows[:200][-30:]
print("busy", sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in a)/3e3, "span/call", (int(a[-1]["End_Timestamp"])-int(a[0]["Start_Timestamp"]))/3e3)
PY
done
