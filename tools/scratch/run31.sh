# leaf hashing: occupancy by launch bounds (5 waves / 87 VGPRs now; 6 and 8 waves cost 112 B of scratch)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for w in 1 6 8; do echo -n "hash waves=$w 116x2^20 | "; tools/scratch/kb/kbench_hw$w 116 20 5 | head -1; done; done
for w in 1 6 8; do echo -n "hash waves=$w 2431x2^17 | "; tools/scratch/kb/kbench_hw$w 2431 17 3 | head -1; done
for w in 1 6 8; do tools/scratch/kb/kbench_hw$w 116 20 1 | tail -1; done
