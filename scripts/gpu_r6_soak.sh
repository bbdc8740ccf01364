D=/dev/shm/nrv_soak_in; O=/dev/shm/nrv_soak_out/
rm -rf $D $O; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5 tests/golden/fast5_more/*.fast5; do
  for k in $(seq 1 10000); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
( while true; do sleep 5; rocm-smi --showmeminfo vram 2>/dev/null | grep "Used" | head -1; ps -o rss= -C python3 | sort -n | tail -1; done ) > /tmp/soak_mon.txt 2>&1 &
MON=$!
python3 NanoReviser.py -d $D -o $O -S human --thread 16 2>&1 | grep -E "bases/s end to end|Error|Warning" | tail -3
kill $MON
ls $O | wc -l
head -1 /tmp/soak_mon.txt; tail -2 /tmp/soak_mon.txt; awk 'NR%2==0' /tmp/soak_mon.txt | sort -n | sed -n '1p;$p'
rm -rf $D $O
