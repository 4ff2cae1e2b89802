ROOT=$(pwd)
for i in 1 2 3; do for lib in base attn4 attn16; do
  echo -n "[$lib] "; STABLETTS_HIP_LIB=$ROOT/tools/ab/$lib.so timeout 200 python tools/class_times.py 2>&1 | tail -1 | cut -c1-75
done; done
