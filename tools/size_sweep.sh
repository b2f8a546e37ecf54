for n in 120000000 150000000 200000000; do
  python bench.py --reads $n --steps 2 --warmup 1 --no-cpu-baseline --no-host-to-host --e2e-reads 0 --no-event-ab --no-check > gpurun_out/size_$n.log 2>&1 && python tools/show_line.py gpurun_out/size_$n.log | head -3
done
