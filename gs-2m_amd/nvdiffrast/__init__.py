"""Import path of the reference's `import nvdiffrast.torch as dr` (pbr/shade.py:8, pbr/light.py:5) on MI355X: only
`dr.texture` in the three modes the reference calls is provided (see nvdiffrast/torch/__init__.py)."""
