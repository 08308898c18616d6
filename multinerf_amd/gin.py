"""A small in-repo stand-in for the subset of gin-config the reference uses.

gin is not installable here (no network); the reference's configs/*.gin only use
`Class.attr = <python literal>`, `@module.fn` references, `include '<path>'`
and `#` comments (reference configs/360.gin:6,9; llff_raw.gin:44;
llff_raw_test.gin:1), and its code only calls
`gin.parse_config_files_and_bindings(files, bindings, skip_unknown=True)`
(internal/configs.py:185-186), `@gin.configurable`, `gin.config_str()` and
`gin.config.external_configurable` (configs.py:29-47, models.py:35-36).

`@module.fn` references resolve to the NAME of the function (the last dotted
component, e.g. '@jnp.reciprocal' -> 'reciprocal'): on this stack the named
functions are HIP kernel modes, not Python callables.
"""

import ast
import dataclasses
import os
import re

_REGISTRY = {}   # configurable name -> class
_BINDINGS = {}   # configurable name -> {attr: value}
_SEARCH_PATHS = ['']

# Functions the reference registers as external configurables
# (configs.py:29-42, models.py:35-36), by gin name.
EXTERNAL_CONFIGURABLES = {
    'jnp.reciprocal': 'reciprocal', 'jnp.log': 'log', 'jnp.log1p': 'log1p',
    'jnp.exp': 'exp', 'jnp.sqrt': 'sqrt', 'jnp.square': 'square',
    'jax.nn.relu': 'relu', 'jax.nn.softplus': 'softplus', 'jax.nn.silu': 'silu',
    'math.safe_exp': 'safe_exp', 'coord.contract': 'contract',
}


class GinError(ValueError):
  pass


def add_config_file_search_path(path):
  _SEARCH_PATHS.append(path)


def clear_config():
  _BINDINGS.clear()


def configurable(cls=None, *, name=None):
  """Class decorator: construction picks up bound values as defaults."""

  def wrap(c):
    gin_name = name or c.__name__
    _REGISTRY[gin_name] = c
    orig_init = c.__init__

    def __init__(self, *args, **kwargs):
      bound = dict(_BINDINGS.get(gin_name, {}))
      bound.update(kwargs)
      orig_init(self, *args, **bound)

    c.__init__ = __init__
    c._gin_name = gin_name
    return c

  return wrap(cls) if cls is not None else wrap


def _resolve_ref(ref):
  ref = ref.lstrip('@').rstrip('()')
  if ref in EXTERNAL_CONFIGURABLES:
    return EXTERNAL_CONFIGURABLES[ref]
  # Tolerate module-path variations ('jax.numpy.reciprocal', ...).
  tail = ref.split('.')[-1]
  for k, v in EXTERNAL_CONFIGURABLES.items():
    if k.split('.')[-1] == tail:
      return v
  raise GinError(f'unknown configurable reference @{ref}')


def _parse_value(text):
  """A Python literal in which `@a.b.c` stands for a configurable reference."""
  refs = []

  def sub(m):
    refs.append(_resolve_ref(m.group(0)))
    return f'"__gin_ref_{len(refs) - 1}__"'

  text2 = re.sub(r'@[A-Za-z_][\w\.]*(\(\))?', sub, text)
  try:
    val = ast.literal_eval(text2)
  except (ValueError, SyntaxError) as e:
    raise GinError(f'cannot parse gin value {text!r}: {e}') from e

  def restore(v):
    if isinstance(v, str):
      m = re.fullmatch(r'__gin_ref_(\d+)__', v)
      return refs[int(m.group(1))] if m else v
    if isinstance(v, (list, tuple)):
      return type(v)(restore(x) for x in v)
    if isinstance(v, dict):
      return {restore(k): restore(x) for k, x in v.items()}
    return v

  return restore(val)


def _strip_comment(line):
  out, quote = [], None
  for ch in line:
    if quote:
      out.append(ch)
      if ch == quote:
        quote = None
    elif ch in '\'"':
      quote = ch
      out.append(ch)
    elif ch == '#':
      break
    else:
      out.append(ch)
  return ''.join(out).rstrip()


def _balanced(s):
  depth, quote = 0, None
  for ch in s:
    if quote:
      if ch == quote:
        quote = None
    elif ch in '\'"':
      quote = ch
    elif ch in '([{':
      depth += 1
    elif ch in ')]}':
      depth -= 1
  return depth <= 0


def _statements(text):
  buf = ''
  for raw in text.splitlines():
    line = _strip_comment(raw)
    if not line.strip() and not buf:
      continue
    buf = (buf + ' ' + line.strip()) if buf else line.strip()
    if buf.endswith('\\'):
      buf = buf[:-1]
      continue
    if _balanced(buf):
      yield buf
      buf = ''
  if buf:
    raise GinError(f'unterminated gin statement: {buf!r}')


def _find_file(path, relative_to=None):
  cands = [path]
  if relative_to:
    cands.append(os.path.join(os.path.dirname(relative_to), path))
  cands += [os.path.join(p, path) for p in _SEARCH_PATHS]
  for c in cands:
    if os.path.isfile(c):
      return c
  raise GinError(f'gin config file not found: {path}')


def parse_config(text, skip_unknown=True, _origin=None):
  for stmt in _statements(text):
    m = re.fullmatch(r"include\s+['\"](.+)['\"]", stmt)
    if m:
      parse_config_file(m.group(1), skip_unknown=skip_unknown, _relative_to=_origin)
      continue
    if '=' not in stmt:
      raise GinError(f'cannot parse gin statement: {stmt!r}')
    lhs, rhs = stmt.split('=', 1)
    lhs = lhs.strip()
    if '.' not in lhs:
      raise GinError(f'gin binding needs Configurable.attr: {stmt!r}')
    target, attr = lhs.rsplit('.', 1)
    target = target.split('/')[-1]      # drop scopes ('train/Config.x')
    target = target.split('.')[-1]      # drop module path
    if target not in _REGISTRY:
      if skip_unknown:
        continue
      raise GinError(f'unknown configurable {target!r}')
    cls = _REGISTRY[target]
    fields = {f.name for f in dataclasses.fields(cls)}
    if attr not in fields:
      raise GinError(f'{target} has no parameter {attr!r}')
    _BINDINGS.setdefault(target, {})[attr] = _parse_value(rhs.strip())


def parse_config_file(path, skip_unknown=True, _relative_to=None):
  full = _find_file(path, _relative_to)
  with open(full) as f:
    parse_config(f.read(), skip_unknown=skip_unknown, _origin=full)


def parse_config_files_and_bindings(config_files, bindings, skip_unknown=True):
  """internal/configs.py:185-186."""
  for f in (config_files or []):
    parse_config_file(f, skip_unknown=skip_unknown)
  for b in (bindings or []):
    parse_config(b, skip_unknown=skip_unknown)


def query_parameter(name):
  target, attr = name.rsplit('.', 1)
  return _BINDINGS[target][attr]


def config_str():
  """The operative bindings, one per line (configs.py:188-191 snapshot)."""
  inv = {v: k for k, v in EXTERNAL_CONFIGURABLES.items()}
  lines = []
  for target in sorted(_BINDINGS):
    for attr in sorted(_BINDINGS[target]):
      v = _BINDINGS[target][attr]
      if isinstance(v, str) and v in inv and attr.endswith(('_fn', '_activation')):
        r = '@' + inv[v]
      else:
        r = repr(v)
      lines.append(f'{target}.{attr} = {r}')
  return '\n'.join(lines) + '\n'
