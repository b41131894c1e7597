a=$(timeout -k 10 300 tools/run_llama_bench.sh 8b Q8_0 -p 2048 -n 128 -ctk q8_0 -ctv q8_0 -r 2 2>&1 | tail -1)
b=$(timeout -k 10 400 tools/run_llama_bench.sh 70b Q4_K_M -p 512 -n 64 -r 2 2>&1 | tail -1)
c=$(timeout -k 10 400 tools/run_llama_bench.sh 8x7b Q4_K_M -p 512 -n 64 -r 2 2>&1 | tail -1)
echo "$a"; echo "$b"; echo "$c"
