for v in x0lb3; do
  RTK_LIB=$PWD/raytracinginoneweekend_amd/variants/librtk_$v.so timeout -k 10 300 python tools/determinism_check.py 3 > gpurun_out/dbg_$v.txt 2>&1
  rc=$?
  cat gpurun_out/dbg_$v.txt | tail -4
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping rc=$rc"; exit 1; fi
done
