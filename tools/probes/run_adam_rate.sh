for g in 2048 32768 65536 131072 1000000; do for nt in 0 2; do CLV_ADAM_NT=$nt CLV_ADAM_GRID=$g python tools/probes/adam_rate.py 2>&1 | grep NT=; done; done
for g in 2048 65536 1000000; do CLV_ADAM_GRID=$g python tools/probes/adam_rate.py 2>&1 | grep NT=; done
