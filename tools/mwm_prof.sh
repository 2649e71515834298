# the blossom kernel's time split on the critical SRtest150 graph (record 217): bash tools/mwm_prof.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
o=gpurun_out/s2; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_MWM_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)" 
python tools/mwm_one.py 217 1 2 > $o/mwm_prof1.txt 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
python tools/mwm_one.py 217 1 3 > $o/mwm_base.txt 2>&1
cat $o/mwm_prof1.txt $o/mwm_base.txt
