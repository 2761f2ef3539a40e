cp fibergen_amd/libfibergen_amd.so /tmp/new.so; cp fibergen_amd/libfibergen_amd_old.so /tmp/old.so
for rep in 1 2; do for v in old new; do
  cp /tmp/$v.so fibergen_amd/libfibergen_amd.so
  for n in 300 400 500 200; do echo -n "$v "; timeout 300 python tools/ab_grid.py --grid $n,$n,$n --steps 10 2>&1 | cut -c1-280; done
done; done
cp /tmp/new.so fibergen_amd/libfibergen_amd.so
