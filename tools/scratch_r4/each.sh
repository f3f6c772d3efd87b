python -m pytest tests/test_gpu_deconv.py -q -m gpu -k "extended_epilogue" --collect-only 2>/dev/null | grep "::" > /tmp/ids.txt
wc -l /tmp/ids.txt
while read id; do
  out=$(timeout 120 python -m pytest "$id" -q -m gpu 2>&1 | grep -E "passed|failed|Abort|fault" | head -1)
  case "$out" in *passed*) ;; *) echo "$id => $out";; esac
done < /tmp/ids.txt
echo done
