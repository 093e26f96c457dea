function results = admm(xminf, zming, options)
%ADMM  Engine-backed replacement of the reference's ADMM loop.
%
%   results = admm(xminf, zming, options)
%
% Same signature, option fields and result fields as the reference (admm.m:24; options admm.m:51-76 with
% the defaults of setopt, admm.m:780-971; results admm.m:257-259, 582-616, 648-656, 681-682, 746-767).
% The loop itself -- x-update, relaxation, z-update, u-update, fast / accelerated ADMM, objective,
% histories, residuals and tolerances, H-norm test, the three stop conditions (admm.m:496-743) -- runs on
% the GPU inside libadmm_hip.so; this file only decides WHAT to hand to the gateway admm_mex:
%
%   xminf / zming are descriptors from getproxops (this directory)  -> the engine's own operators
%   xminf / zming are MATLAB function handles f(x, z, u, rho)        -> staged through host memory once per
%       iteration by the gateway (mexCallMATLAB); everything else stays on the device.  Mixed pairs are
%       fine: unwrappedadmm.m:78 builds xminf = @(x,z,u,rho) Dplus*(z-u) next to the library's SVM
%       z-update (that particular handle is recognised and runs natively, with its Dplus);
%       examples/convergencechecking.m:125-136 mixes a library operator with a deliberately broken one.
%   options.obj is the closure a reference solver installs (lasso.m:227, lad.m:148, huberfit.m:180,
%       totalvariation.m:134-135, linearsvm.m:232-236, basispursuit.m:140, linearprogram.m:180,
%       quadraticprogram.m:242)                                      -> the engine's own objective;
%       any other handle                                             -> evaluated by MATLAB once per iteration.
%
%   options.altu / options.specialnorms as the caller's handles (admm.m:553-559, 612-616) -> staged like the prox
%       handles (plain ADMM only); options.preprocess (admm.m:473-476) is called here, once, before the loop;
%   options.adaptive with options.convtest (experimental in the reference, admm.m:724-741) -> stepped by this file:
%       one device iteration per gateway call on a persistent engine, rho updated by the reference's rule, the
%       engine re-created for a new rho where the reference's closure re-factors (adaptiveloop below).

if ~isstruct(options)
    error('Given options is not a struct! At least pass empty struct!');    % admm.m:48
end

xd = isdescriptor(xminf);
zd = isdescriptor(zming);
handles = struct();

adaptive = isfield(options, 'adaptive') && any(options.adaptive) && isfield(options, 'convtest') && any(options.convtest);

if xd && zd
    if xminf.id ~= zming.id
        error('admm:mixed', 'The two library operators come from different getproxops calls.');
    end
    problem = xminf.problem;
    args = xminf.args;
elseif zd           % library z-update, caller's x-update (unwrappedadmm.m:78)
    problem = zming.problem;
    args = zming.args;
    [native, args] = unwrappedxupdate(xminf, problem, args);
    if ~native
        handles.xminf = slicewrap(xminf, options, 'xminf');
    end
elseif xd           % library x-update, caller's z-update (linearprogram.m:158-164 altproxg, quadraticprogram.m)
    problem = xminf.problem;
    args = xminf.args;
    handles.zming = slicewrap(zming, options, 'zming');
else                % both operators are the caller's: the generic loop (admm.m:24)
    problem = 'generic';
    [args, handles] = genericargs(options, handles);
    handles.xminf = slicewrap(xminf, options, 'xminf');
    handles.zming = slicewrap(zming, options, 'zming');
end

if xd || zd
    checkconstraint(problem, options);
end

% options.altu / options.specialnorms: the consensus hooks are descriptors and stay the engine's own (lasso.m:222-223
% copies them from extra); a caller's function handle travels to the gateway with the other handles
hooks = {'altu', 'specialnorms'};
for k = 1:numel(hooks)
    if isfield(options, hooks{k}) && isa(options.(hooks{k}), 'function_handle')
        handles.(hooks{k}) = options.(hooks{k});
    end
end
if isfield(options, 'preprocess') && isa(options.preprocess, 'function_handle')      % admm.m:473-476
    options.preprocess();
end

[args, handles] = attachobjective(problem, args, options, handles);

% rho the cached factor is built for = the rho of this run (the reference's closures re-factor on
% rho ~= rhoprev, getProxOps.m:1222-1238, 1446-1453; xminLASSO keeps the caller's factor, 1192-1206)
% (the linear program's args carry no rho at all: the engine eliminates its KKT multiplier for this rho, on the device)
if isfield(options, 'rho') && ~(isfield(args, 'L') || isfield(args, 'R'))
    args.rho = options.rho;
end

if adaptive
    results = adaptiveloop(problem, args, options, handles);
else
    results = admm_mex('solve', problem, args, options, handles);
end
results.options = options;                                                  % admm.m:767

end

% -------------------------------------------------------------------------------------------------------
function results = adaptiveloop(problem, args, options, handles)
% admm.m:724-741 (experimental): after every iteration i > 2, rho <- rho*rhoprev/(H1 - H2), clamped to a factor of 5 per
% step.  One device iteration per 'run' on a persistent engine, warm-started from the previous one; closures with a
% "rho ~= rhoprev" branch (getProxOps.m:1222-1238, 1446-1453; the KKT solves 1363, 1410) get a new engine for a new rho,
% xminLASSO / xminLAD keep their factor (1192-1206, 1511-1515).  tests/test_mex_gateway.py steps the gateway the same way.
refactor = any(strcmp(problem, {'quadraticprogram', 'linearprogram', 'model'})) || ...
    (isfield(args, 'parallel') && any(args.parallel));
rho = 1.0;
if isfield(options, 'rho'), rho = options.rho; end
rhoH = rho;                                                                 % admm.m:305-309 captures rho once
N = 1000;
if isfield(options, 'maxiters') && options.maxiters > 0, N = ceil(options.maxiters); end
step = rmfield(options, intersect(fieldnames(options), {'adaptive', 'preprocess'}));
step.convtest = 0; step.stopcond = 'standard'; step.maxiters = 1; step.domaxiters = 1; step.recordhistory = 0;
if ~refactor, step.stalefactorok = 1; end
h = []; rhobuilt = NaN; results = struct(); H = zeros(1, 0); w = []; runtime = 0;
cleanup = onCleanup(@() destroyengine(h));
for i = 1:N
    if isempty(h) || (refactor && rho ~= rhobuilt)
        destroyengine(h);
        args.rho = rho;
        h = admm_mex('create', problem, args, handles);
        cleanup = onCleanup(@() destroyengine(h)); %#ok<NASGU>
        rhobuilt = rho;
    end
    step.rho = rho;
    r = admm_mex('run', h, step, handles);
    if i == 1
        results.x0 = r.x0; results.z0 = r.z0; results.u0 = r.u0;
        w = [r.x0; r.z0; rho*r.u0];
    end
    step.x0 = r.xopt; step.z0 = r.zopt; step.u0 = r.uopt;
    runtime = runtime + r.runtime;
    results.xvals(:, i) = r.xopt; results.zvals(:, i) = r.zopt; results.uvals(:, i) = r.uopt;
    results.pnorm(i) = r.pnorm; results.dnorm(i) = r.dnorm; results.perr(i) = r.perr; results.derr(i) = r.derr;
    if isfield(r, 'objevals'), results.objevals(i) = r.objevals; end
    wprev = w; w = [r.xopt; r.zopt; rho*r.uopt];
    results.wvals(:, i) = w;
    nz = numel(r.zopt); nx = numel(r.xopt);
    dw = wprev - w;
    H(i) = rhoH*norm(dw(nx+1:nx+nz))^2 + rhoH*norm(dw(nx+nz+1:end))^2;      % admm.m:305-309, 681-682
    results.Hnormsq(i) = H(i);
    stopnow = ~(isfield(options, 'domaxiters') && any(options.domaxiters)) && r.pnorm < r.perr && ...
        ((isfield(options, 'nodualerror') && any(options.nodualerror)) || r.dnorm < r.derr);
    if stopnow, break; end
    if i > 2                                                                 % admm.m:724-741
        wdiff = H(i-1) - H(i);
        rhoprev = rho;
        rho = rho*(wdiff*rhoprev)/(wdiff*wdiff);
        if abs(rho - rhoprev) >= rhoprev*5
            rho = rho/5;
        elseif abs(rho - rhoprev) <= rhoprev/5
            rho = rho*5;
        end
    end
end
results.steps = i; results.xopt = r.xopt; results.zopt = r.zopt; results.uopt = r.uopt;
if isfield(r, 'objopt'), results.objopt = r.objopt; end
results.runtime = runtime;
end

function destroyengine(h)
if ~isempty(h)
    try
        admm_mex('destroy', h);
    catch
    end
end
end

% -------------------------------------------------------------------------------------------------------
function tf = isdescriptor(f)
tf = isstruct(f) && isfield(f, 'admm_engine_descriptor');
end

function checkconstraint(problem, options)
% the library operators fix the constraint of their problem (lasso.m:232-238, lad.m:140-145,
% totalvariation.m:151-157, unwrappedadmm.m:81-86): function-handle A / At / B or a B other than -1 only go
% with two caller-supplied handles (the generic loop)
names = {'A', 'At', 'B'};
for k = 1:numel(names)
    if isfield(options, names{k}) && isa(options.(names{k}), 'function_handle')
        error('admm:unsupported', ['options.', names{k}, ' as a function handle needs both proximal operators ', ...
            'to be the caller''s (', problem, ').']);
    end
end
if isfield(options, 'B') && isnumeric(options.B) && ~(isscalar(options.B) && options.B == -1)
    error('admm:unsupported', 'The library operators are written for B = -1 (every reference solver uses it).');
end
end

function [args, handles] = genericargs(options, handles)
% admm.m:79-245: c, A / At and B of the constraint A*x + B*z = c.  Matrices and scalars travel in args, function
% handles in handles (the gateway calls them back with host vectors); the engine runs the loop on the device.
args = struct();
if ~isfield(options, 'A')
    error('Must specify a matrix A in constraint Ax + Bz = c!');                 % admm.m:155-157
end
if ~isfield(options, 'B')
    error('Must specify a matrix B in constraint Ax + Bz = c!');                 % admm.m:241-244
end
m = 0;
if isfield(options, 'm'), m = options.m; end
if isfield(options, 'c') && isnumeric(options.c) && numel(options.c) > 1
    m = numel(options.c);
end
A = options.A;
if isa(A, 'function_handle')                                                     % admm.m:121-130, 165-178
    if ~isfield(options, 'nA') || options.nA == 0
        error(['Matrix A is a function handle, but no number of columns nA specified for it; cannot infer nA - ', ...
            'please specify it in options struct!']);
    end
    if ~isfield(options, 'At') || ~isa(options.At, 'function_handle')
        error('admm:arg', 'options.A is a function handle: options.At must be one too.');
    end
    if m == 0, m = numel(A(zeros(options.nA, 1))); end
    handles.A = A;
    handles.At = options.At;
    args.nA = options.nA;
    args.m = m;
elseif isnumeric(A) && numel(A) > 1
    args.A = full(A);
    m = size(A, 1);
elseif isnumeric(A) && isscalar(A) && A ~= 1                                     % a*I as the pair v -> a*v
    if m == 0 && isfield(options, 'nA'), m = options.nA; end
    handles.A = @(v) A*v;
    handles.At = handles.A;
    args.nA = m;
    args.m = m;
else
    if isfield(options, 'nA') && options.nA > 0
        args.n = options.nA;
    elseif m > 0
        args.n = m;
    elseif isfield(options, 'x0')
        args.n = numel(options.x0);
    else
        error('admm:arg', 'The generic loop needs the vector length: set options.nA (or m, or x0).');
    end
    m = args.n;
end
B = options.B;
if isa(B, 'function_handle')                                                     % admm.m:206-216
    if ~isfield(options, 'nB') || options.nB == 0
        error(['Matrix B is a function handle, but no number of columns nB specified for it; cannot infer nB - ', ...
            'please specify it in options struct!']);
    end
    handles.B = B;
    args.nB = options.nB;
elseif isnumeric(B) && numel(B) > 1
    args.B = full(B);
elseif isnumeric(B) && isscalar(B)
    args.B = B;
else
    error(['Given B in constraint Ax + Bz = c is neither a numeric matrix nor function handle of single vector!']);
end
if isfield(options, 'c') && isnumeric(options.c)
    if isscalar(options.c)
        args.c = options.c*ones(m, 1);                                          % admm.m:79-110
    else
        args.c = options.c(:);
    end
end
end

function [native, args] = unwrappedxupdate(xminf, problem, args)
% unwrappedadmm.m:76-78  xminf = @(x,z,u,rho) Dplus*(z - u)   -> the engine applies Dplus itself
% unwrappedadmm.m:125-141 xminf = @proxf (transpose reduction) -> the engine's cached factor of sum Di'*Di
native = false;
if ~strcmp(problem, 'linearsvm') || ~isa(xminf, 'function_handle')
    return;
end
txt = regexprep(func2str(xminf), '\s', '');
if strcmp(txt, '@(x,z,u,rho)Dplus*(z-u)')
    native = true;
    if ~isfield(args, 'Dplus')
        info = functions(xminf);
        if isfield(info, 'workspace') && ~isempty(info.workspace) && isfield(info.workspace{1}, 'Dplus')
            args.Dplus = info.workspace{1}.Dplus;
        end
    end
elseif ~isempty(regexp(txt, 'unwrappedadmm/proxf$', 'once'))
    native = true;
end
end

function f = slicewrap(f, options, which)
% admm.m:343-468 (parproxf / parproxg): with options.parallel naming this operator, the handle has the
% form f(x, z, u, rho, k) and returns slice k; the slices are concatenated in order
if ~isa(f, 'function_handle')
    error('admm:arg', 'A proximal operator must be a function handle or a getproxops descriptor.');
end
if ~isfield(options, 'parallel') || ~ischar(options.parallel)
    return;
end
if ~(strcmp(options.parallel, which) || strcmp(options.parallel, 'both'))
    return;
end
if ~isfield(options, 'slices')
    error('admm:arg', 'options.parallel needs options.slices.');
end
count = numel(options.slices);
g = f;
f = @(x, z, u, rho) concatslices(g, x, z, u, rho, count);
end

function v = concatslices(g, x, z, u, rho, count)
parts = cell(count, 1);
for k = 1:count
    parts{k} = reshape(g(x, z, u, rho, k), [], 1);
end
v = cell2mat(parts);
end

function [args, handles] = attachobjective(problem, args, options, handles)
% options.obj (admm.m:248, 603-605).  The closures the reference solvers install are restated on the device;
% they are recognised by their text.  Anything else is evaluated by MATLAB.
if ~(isfield(options, 'objevals') && any(options.objevals)) || ~isfield(options, 'obj') || ...
        ~isa(options.obj, 'function_handle')
    return;
end
txt = regexprep(func2str(options.obj), '\s', '');
known = struct( ...
    'lasso', {{'@(x,z)0.5*sum((D*x-s).^2)+lambda*norm(z,1)'}}, ...
    'lad', {{'@(x,z)norm(z,1)'}}, ...
    'huberfit', {{'@(x,z)1/2*sum(huber(z))'}}, ...
    'totalvariation', {{'@(x,z)1/2*norm(x-s,''fro'')^2+lambda*sum(abs(x(2:length(x))-x(1:length(x)-1)))'}}, ...
    'linearsvm', {{'@(x,z)1/2*norm(x,''fro'')^2+C*sum(max(1-ell.*(D*x),0))', ...
                   '@(x,z)1/2*norm(x,''fro'')^2+C*sum(max(sign(1-ell.*(D*x)),0))'}}, ...
    'basispursuit', {{'@(x,z)norm(x,1)'}}, ...
    'linearprogram', {{'@(x,z)b''*x'}}, ...
    'quadraticprogram', {{'@(x,z)1/2*x''*P*x+q''*x+r'}});
if isfield(known, problem) && any(strcmp(txt, known.(problem)))
    handles.objnative = 1;
    info = functions(options.obj);
    ws = struct();
    if isfield(info, 'workspace') && ~isempty(info.workspace)
        ws = info.workspace{1};
    end
    if strcmp(problem, 'lasso') && ~isfield(args, 's')                     % getProxOps.m:445-451 has no s
        if isfield(ws, 's')
            handles.s = ws.s;
        else
            handles = rmfield(handles, 'objnative');
            handles.obj = options.obj;
        end
    end
    if strcmp(problem, 'quadraticprogram') && isfield(ws, 'r')              % quadraticprogram.m:242
        handles.r = ws.r;
    end
else
    handles.obj = options.obj;
end
end
