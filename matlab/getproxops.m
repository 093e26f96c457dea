function [minx, minz, extra] = getproxops(problem, args)
%GETPROXOPS  Engine-backed replacement of the reference's prox-operator factory.
%
%   [minx, minz, extra] = getproxops(problem, args)
%
% Same name, arguments and outputs as the reference's function (file getProxOps.m, function getproxops,
% line 13; every caller spells it lower case: lasso.m:192,221; lad.m:137; huberfit.m:169;
% linearsvm.m:217; totalvariation.m:148; quadraticprogram.m:228-232; linearprogram.m:166-170;
% basispursuit.m:127; model.m:124).  Put this directory BEFORE the reference's root on the MATLAB path
% (addpath(..., '-begin')); the solver files then run unchanged.
%
% Instead of nested-function closures it returns DESCRIPTORS: structs that record the problem string and
% the args struct.  admm.m (same directory) recognises them and runs the whole iteration on the GPU
% through admm_mex (admm-project_amd/csrc/admm_mex.cpp -> libadmm_hip.so).  A descriptor is not callable:
% the operators only ever run inside the engine.  Nothing is uploaded here -- the engine is created by
% admm(), which also sees options (rho, the objective handle, the constraint data).
%
% Problems: 'model', 'basispursuit', 'totalvariation', 'linearsvm', 'lasso' (serial and args.parallel),
% 'linearprogram', 'quadraticprogram' (args.constraint 'bounded' or 'standard'), 'lad', 'huberfit', and the
% engine-side extension 'totalvariation2d' (args.S = image, args.lambda).  'covarianceselection' is not
% engine-native (eigen-decomposition bound; keep the reference's closures for it).

persistent counter
if isempty(counter)
    counter = 0;
end

extra = struct();

if ischar(problem)
    problem = lower(problem);
else
    error(['Given problem argument is not a string specifying for ', ...
        'which problem proximal operators are needed!']);
end

if ~isstruct(args)
    error(['Given struct args is not a struct containing arguments ', ...
        'needed for proximal operators for the given problem!']);
end

native = {'model', 'basispursuit', 'totalvariation', 'linearsvm', 'lasso', 'linearprogram', ...
    'quadraticprogram', 'lad', 'huberfit', 'totalvariation2d'};

if strcmp(problem, 'covarianceselection')
    error('admm:unsupported', ['covarianceselection is not engine-native (symmetric eig per iteration); ', ...
        'use the reference''s own getproxops for it.']);
elseif ~ismember(problem, native)
    error('Invalid input for problem - given string is not a solver!');
end

if ~admm_mex('available')
    error('admm:nodevice', 'No HIP device is visible: the ADMM engine has no CPU fallback.');
end

counter = counter + 1;
d = struct('admm_engine_descriptor', true, 'id', counter, 'problem', problem, 'args', args, 'role', 'x');
minx = d;
minz = d;
minz.role = 'z';

% getProxOps.m:663-666: a caller-supplied conic projection replaces the z-update of the QP
if strcmp(problem, 'quadraticprogram') && isfield(args, 'altproxg') && ...
        isa(args.altproxg, 'function_handle')
    minz = args.altproxg;
end

% getProxOps.m:441-442: consensus lasso hands back its u-update and its norms
if strcmp(problem, 'lasso') && isfield(args, 'parallel') && any(args.parallel)
    extra.altu = d;
    extra.altu.role = 'altu';
    extra.specialnorms = d;
    extra.specialnorms.role = 'specialnorms';
end

end
