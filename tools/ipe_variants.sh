export MNR_SKIP_PREFLIGHT=1
python tools/ipe_probe.py
for v in fe1 fe2 fe3 fe4 ipea ipeb ipec iped ipee ipef; do MNR_LIB_PATH=$PWD/multinerf_amd/libmnerf_hip_$v.so timeout 60 python tools/ipe_probe.py 2>&1 | tail -n 2; done
python tools/ipe_probe.py
