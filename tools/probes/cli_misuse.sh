printf '2 2 650 865\n10 0 30 20\n60 0 50 180\n' > /tmp/in.txt
for env in "GORTT_DEVICES=5" "GORTT_DEVICES=0,7" "GORTT_DEVICES=abc" "GORTT_DEVICES=-1" "GORTT_CHUNK_MB=0" "GORTT_CHUNK_MB=-5" "GORTT_CHUNK_MB=100000" "GORTT_THREADS=0" "GORTT_THREADS=1000"; do
  echo "== $env"; env $env timeout 60 ./gort_amd/bin/gortt -LAI 4.0 < /tmp/in.txt 2>&1 | tail -3; echo "rc=${PIPESTATUS[0]}"
done
for a in "--gpus 0" "--gpus 9" "--gpus -1" "--gpus x" "--gpus" "--lut-cache" "--binary-in" "--bogus"; do
  echo "== $a"; timeout 60 ./gort_amd/bin/gortt -LAI 4.0 $a < /tmp/in.txt 2>&1 | tail -2 | cut -c1-200; echo "rc=${PIPESTATUS[0]}"
done
