import sys, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
print(json.dumps(bench._sec_joint()))
