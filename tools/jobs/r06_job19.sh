cp fibergen_amd/libfibergen_amd.so /tmp/new.so; cp fibergen_amd/libfibergen_amd_old.so /tmp/old.so
for rep in 1 2; do for v in old new; do
  cp /tmp/$v.so fibergen_amd/libfibergen_amd.so
  for g in 32,256,256 64,512,512 256,256,256 512,512,512 160,160,160; do echo -n "$v "; timeout 300 python tools/ab_grid.py --grid $g --steps 20 --set plane_fft=0 2>&1 | cut -c1-300; done
done; done
cp /tmp/new.so fibergen_amd/libfibergen_amd.so
