set -x
python tests/tools/gpu_parity.py --kind kitti --frames 60 --seed 1 --verbose 0 --planes 0 2>&1 | tail -15
python tests/tools/gpu_parity.py --kind kitti --frames 15 --seed 2 --verbose 0 --set agast_threshold=12 --set max_keypoints_per_cell=60 2>&1 | tail -15
python tests/tools/gpu_parity.py --kind kitti --frames 20 --seed 3 --verbose 1 --planes 0 --jump 8 2>&1 | tail -25
python tests/tools/gpu_parity.py --kind kitti --frames 8 --seed 4 --verbose 1 --set agast_threshold=150 2>&1 | tail -15
python tests/tools/gpu_parity.py --kind euroc --frames 30 --verbose 0 2>&1 | tail -15
python tests/tools/gpu_parity.py --kind tum --frames 30 --verbose 0 2>&1 | tail -15
python tests/tools/gpu_parity.py --kind kitti --frames 30 --seed 5 --verbose 0 --planes 0 --set triangulation_policy=2 --set staged_threshold=0 2>&1 | tail -8
