"""Ray / bounding-volume intersection operators (reference: python/intersection/)."""
