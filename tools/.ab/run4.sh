cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_data_pipeline.py -x -q -m gpu -k "graphed or trainer" 2>&1 | grep -v "^  File \"/usr\|Extension modules\|amdgpu.ids" | tail -25
