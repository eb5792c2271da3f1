"""find_amd -- MI355X-native hot path of FIND (per-vertex implicit deformation/colour MLP, differentiable mesh
render, Chamfer / smoothness losses) behind FIND's own Python model API.  See DESIGN.md."""
__version__ = '0.1.0'
