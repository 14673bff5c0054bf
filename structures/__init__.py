"""`import structures` == `import monoflex_amd.structures` (the reference's top-level name, tools/plain_train_net.py:9-26)."""
from monoflex_amd._alias import install

install(__name__)
