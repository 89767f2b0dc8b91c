#!/bin/bash
# round 5, GPU pass am: the driver's smoke() on the final build
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
