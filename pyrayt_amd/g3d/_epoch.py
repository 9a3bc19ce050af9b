"""The change counter of the scene objects (its own module: world objects and materials both count)."""


class SceneEpoch:
    """A process-wide counter that moves whenever an attribute of any scene object is assigned -- a transform applied
    (``_append_world_transform`` assigns the matrices), a material or a normal sign changed, a wavelength set, a
    material's own numbers (``Material.__setattr__``: a refractive index, a Sellmeier coefficient), a shape's parameters.  What
    holds a compiled copy of a scene (``RayTracer``: the device scene of its components, the ray set of its sources)
    remembers the value it was compiled at and looks at the objects again only when it has moved: a design loop that
    calls ``trace()`` on an unchanged system pays nothing for it, one that moved a part re-snapshots as ever.  Not
    seen: arrays edited in place (``part._world[0, 3] += 1`` -- upstream's own caches, the inverse matrix and the CSG
    cull boxes, would be stale as well) and the insides of a user's material object (such systems are looked at
    again on every trace); ``RayTracer.invalidate()`` is the way out for anything of that kind."""

    value = 0
