import json, os
print(json.dumps(dict(os.environ)))
