for n in 3 4 6 8 16; do echo "FWD_NBY=$n"; DS_FWD_NBY=$n timeout -k 10 60 scripts/kbench depth_estimation_amd/libdepthsynth.so 12 2 192 640 200 | grep -E "wall clock" || exit 1; done
for n in 4 8 ; do echo "hires FWD_NBY=$n"; DS_FWD_NBY=$n timeout -k 10 60 scripts/kbench depth_estimation_amd/libdepthsynth.so 64 2 320 1024 20 | grep -E "wall clock" || exit 1; done
