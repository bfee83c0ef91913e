O=gpurun_out/r03e; mkdir -p $O
for k in 0 1; do for s in "8192 32 256 continuous" "4096 32 256 continuous" "4096 8 256 discrete9"; do
  echo "KCO=$k" >> $O/probe.log
  UAMENV_MANY3_KCO=$k UAMENV_DEBUG=1 timeout -k 10 60 python tools/many_probe.py $s 30 2>&1 | grep -v "amdgpu.ids\|step kernel" >> $O/probe.log
done; done
cat $O/probe.log
