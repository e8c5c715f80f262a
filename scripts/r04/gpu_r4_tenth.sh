# Round 4, GPU call 10 (the last two minutes): does the compact variant's throughput follow the seeds in flight? Fewer workgroups per CU
# (3 and 4 instead of 5) through lcb_device_opts.compact_slots - if a pass gets slower in proportion, more than 5 would pay.
mkdir -p gpurun_out/r4j
timeout 115 python scripts/ab_engine.py --workload ecoli62 base slots4:dev.compact_slots=1024 slots3:dev.compact_slots=768 > gpurun_out/r4j/ab_slots.txt 2> gpurun_out/r4j/ab_slots.err; cat gpurun_out/r4j/ab_slots.txt
